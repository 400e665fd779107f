"""Regression test for the cross-kernel disturbance of round 4: a precision-2 front-end kernel with an
f16 MFMA fed from LDS (tools/withdrawn/qv_ort_conv0_mfma.hip), running in ANOTHER engine, made the log-mel kernel compute a
few wrong power-spectrum bins.  tools/interference_probe recreates the situation without the engine: stream V recomputes
the log-mel features of one ragged batch over and over through launch_logmel -- i.e. through whichever kernel the product
ships -- and counts values that differ from its first, undisturbed result, while stream A runs the aggressor.

  * the shipped log-mel kernel next to the withdrawn aggressor (range pass, the worst case seen): 0 differing values;
  * the shipped log-mel kernel next to the precision-2 front end that ships (conv.0 on the f32 matrix pipe): 0;
  * every k_logmel variant of the probe reproduces the shipped kernel bit for bit when nothing else runs.
Any new kernel that shares the chip with k_logmel belongs in the probe's aggressor list.

Round 5 found the victim instruction: packed FP32 (v_pk_add/mul/fma_f32).  With the FFT output and every later stage dumped
per frame (victims 12 / 13), the FFT output is always right and the first wrong values are the float2 arithmetic of the
real-transform unpack, always in lanes 48-63; compiled without packed-FP32 instructions every variant of the kernel --
LDS exchange, register FFT, block barriers -- is undisturbed, compiled with them every one is disturbed, on two different
boxes (profiles/r05_a_interference_*.log).  The library is therefore built without them (offline-tarteel_amd/build.py), the
probe binaries `current` and `withdrawn` with the library's flags, and `withdrawn_pk` (packed FP32 left on) is the positive
control that shows whether the chip / driver under test still has the hazard at all.
"""

import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def probes():
    sys.path.insert(0, str(ROOT / "tools"))
    import build_probe

    cur, wd, wd_pk = build_probe.build()
    return {"current": cur, "withdrawn": wd, "withdrawn_pk": wd_pk}


def _run(binary, iters, aggr, victim):
    r = subprocess.run([str(binary), str(iters), str(aggr), str(victim)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout)
    m = re.search(r"(\d+) differing values, seen in (\d+) of (\d+) checked groups", r.stdout)
    assert m, r.stdout
    same = re.search(r"nothing else running: (\d+) differing values", r.stdout)
    return int(m.group(1)), int(m.group(2)), (int(same.group(1)) if same else None)


def test_shipped_logmel_is_undisturbed_by_the_withdrawn_kernel(probes):
    """victim 0 = launch_logmel as the product launches it; aggressor 1 = the withdrawn kernel's range pass, 3 = both passes."""
    for aggr in (1, 3):
        diff, groups, _ = _run(probes["withdrawn"], 600, aggr, 0)
        assert diff == 0 and groups == 0, (aggr, diff, groups)


def test_positive_control_packed_fp32_victims(probes):
    """not an assertion about the product: the victims compiled WITH packed FP32 next to the withdrawn aggressor.  On the
    MI355X boxes of round 5 this reports ~1.2 M differing values; 0 here would mean the chip / firmware no longer has the
    hazard.  Either outcome is printed, neither fails."""
    diff, groups, _ = _run(probes["withdrawn_pk"], 200, 1, 6)
    print(f"[interference] positive control (packed FP32 on): {diff} differing values in {groups} groups")


def test_shipped_logmel_is_undisturbed_by_the_shipped_precision2_front_end(probes):
    diff, groups, _ = _run(probes["current"], 600, 3, 0)
    assert diff == 0 and groups == 0, (diff, groups)


@pytest.mark.parametrize("victim", [20, 21, 22, 23, 24, 25])
def test_gemm_kernels_are_undisturbed(probes, victim):
    """Round 6: the GEMM translation units are built without packed-FP32 instructions like the rest of the library; the two
    instantiations that keep them (k_gemm_pk: W8A16 on 128-wide tiles with the GLU / residual epilogue -- they spill without
    the packed forms, csrc/qv_gemm.hip) are victims 24 / 25.  Victims 20-23 are the FFN-up and the long-K GEMM on 256 x 256
    and on 128-wide tiles.  Next to the withdrawn aggressor every output word must stay identical over 400 launches, in the
    product's flavour AND with packed FP32 left on everywhere (`_pk`, how rounds 1-5 compiled the GEMM units)."""
    for flavour in ("withdrawn", "withdrawn_pk"):
        for aggr in (1, 3):
            diff, groups, _ = _run(probes[flavour], 400, aggr, victim)
            assert diff == 0 and groups == 0, (flavour, victim, aggr, diff, groups)


@pytest.mark.parametrize("victim", [6, 7, 10, 11])
def test_logmel_variants_reproduce_the_shipped_kernel(probes, victim):
    """6 = a plain copy, 7 = FFT in registers (csrc/qv_logmel_reg.h), 10 = one frame per block with __syncthreads(),
    11 = split real / imaginary arrays: identical bits on the probe's ragged 30 s batch, nothing else running."""
    diff, _, same = _run(probes["current"], 8, 0, victim)
    assert same == 0 and diff == 0, (victim, same, diff)
