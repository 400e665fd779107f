"""Build libqverse.so (HIP, gfx950 only) in-tree with explicit hipcc commands.

    python offline-tarteel_amd/build.py [--force]

One object per translation unit (compiled in parallel), one link.  -ffp-contract=off:
the retrieval scores are compared bit-for-bit with Python doubles, and nothing on this
path is FMA-throughput bound outside the MFMA kernels (which use intrinsics).
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "build"
LIB = PKG / "libqverse.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No packed-FP32 instructions (v_pk_add/mul/fma_f32) anywhere in the library.  Round 5 traced the "k_logmel computes a few
# wrong power-spectrum bins while a kernel of another engine runs" disturbance of round 4 to them: next to a wave that
# issues v_mfma_f32_32x32x16_f16 fed from LDS at full rate on the same SIMD, a v_pk_*_f32 of the victim wave delivers wrong
# results in lanes 48-63 (tools/interference_probe, victims 12/13: the FFT output is right, the first wrong values are the
# float2 arithmetic of the real-transform unpack, always the last lane quarter; with this switch every victim variant is
# undisturbed, profiles/r05_a_interference_*.log).  Scalar v_mul/add/fma_f32 form the same IEEE results, so no bit changes,
# and the library's kernels are memory- or matrix-pipe-bound: the bench lines do not move (profiles/r05_b_*).
# Round 6: the two GEMM translation units are built this way too (rounds 5 kept the packed forms there).  The ONLY kernels
# that still contain v_pk_*_f32 are two instantiations of the 128-wide GEMM that spill without them (k_gemm_pk<...> in
# csrc/qv_gemm.hip, which switch the feature back on for themselves with a target attribute); tests/test_capi_load.py
# disassembles the library and holds it to exactly that.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
PACKED_F32_TUS: set = set()
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def sources():
    return sorted(CSRC.glob("*.hip"))


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False, dev_hooks: bool = False) -> Path:
    """dev_hooks: compile the timing experiments (QVERSE_SKIP / QVERSE_DUP: drop or duplicate kernel classes) into
    the forward schedule -- never in the product build."""
    OBJ.mkdir(exist_ok=True)
    # the objects of a --dev-hooks build must never be linked into a product build (and vice versa): a flavour stamp
    # forces a full rebuild when the flavour changes (tests/test_capi_load.py also checks the library for the hook names)
    stamp = OBJ / "flavour.txt"
    import hashlib
    flavour = ("dev-hooks" if dev_hooks else "product") + " flags " + hashlib.sha1(
        repr((FLAGS, NO_PACKED_F32, sorted(PACKED_F32_TUS))).encode()).hexdigest()[:12]   # a flag change rebuilds everything
    if not stamp.exists() or stamp.read_text().strip() != flavour:
        force = True
    headers = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + [PKG.parent / "include" / "qverse.h"]
    jobs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC, *FLAGS, *([] if src.stem in PACKED_F32_TUS else NO_PACKED_F32), *(["-DQV_DEV_HOOKS"] if dev_hooks else []),
               "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    stamp.write_text(flavour + "\n")
    objs = [OBJ / (s.stem + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    hooks = "--dev-hooks" in sys.argv
    print(build(force="--force" in sys.argv or hooks, verbose=True, dev_hooks=hooks))
