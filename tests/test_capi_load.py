"""CPU-side checks of the drop-in boundary: the HIP library builds, loads and exports every
symbol include/qverse.h declares (no compute calls without a GPU)."""

import ctypes

import pytest


def test_library_builds_loads_and_exports_header_symbols():
    import importlib.util
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("_qv_build", str(root / "offline-tarteel_amd" / "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    lib_path = b.build()
    lib = ctypes.CDLL(str(lib_path))
    from offline_tarteel_amd.engine import exported_symbols

    syms = exported_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), f"libqverse.so lacks {s}"
    lib.qv_build_info.restype = ctypes.c_char_p
    assert b"gfx950" in lib.qv_build_info()
    lib.qv_frames_for_samples.argtypes = [ctypes.c_int64]
    assert lib.qv_frames_for_samples(160000) == 126
    assert lib.qv_frames_for_samples(80000) == 63
    assert lib.qv_frames_for_samples(480000) == 376


def test_int4_packing_roundtrip_equals_oracle_quantiser():
    """host half of the W4A16 path: quantise + device packing + unpacking == the oracle's
    quantise->dequantise, bit for bit (zeros, ties, constant blocks and outliers included)."""
    import numpy as np
    from pathlib import Path

    from oracle.fastconformer_ref import quant_dequant_int4

    lib = ctypes.CDLL(str(Path(__file__).resolve().parent.parent / "offline-tarteel_amd" / "libqverse.so"))
    lib.qv_debug_int4_roundtrip.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    rng = np.random.default_rng(5)
    for N, K in ((64, 128), (128, 512), (192, 2048)):
        w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        w[0, :128] = 0.0                    # all-zero block: scale 0
        w[1, :128] = 0.25                   # constant block
        w[2, 5] = 3.0                       # positive outlier sets the (negative) scale
        w[3, 7] = -3.0
        w[4, :128] = np.float32(0.1) * np.tile(np.array([1, -1], np.float32), 64)  # +-tie for the extreme
        out = np.empty_like(w)
        rc = lib.qv_debug_int4_roundtrip(w.ctypes.data, N, K, out.ctypes.data)
        assert rc == 0
        ref = quant_dequant_int4(w)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (N, K)
        # half a step everywhere, a whole step where +max clips at code 15
        assert np.abs(out - w).max() <= np.abs(w).max() / 8 * 1.01 + 1e-4
    assert lib.qv_debug_int4_roundtrip(w.ctypes.data, 60, 128, out.ctypes.data) != 0


def test_int8_packing_roundtrip_equals_oracle_quantiser():
    """qv_pack_w8 + the inverse of the device tile layout (host-only) against the oracle's per-channel
    int8 quantiser: bit-identical dequantised matrices."""
    from pathlib import Path

    import numpy as np

    from oracle.fastconformer_ref import quant_dequant_int8

    lib = ctypes.CDLL(str(Path(__file__).resolve().parent.parent / "offline-tarteel_amd" / "libqverse.so"))
    lib.qv_debug_int8_roundtrip.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    rng = np.random.default_rng(6)
    for N, K in ((64, 64), (128, 512), (1024, 512)):
        w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        w[0] = 0.0                          # all-zero row: scale 1
        w[1] = 0.25
        w[2, 5] = 3.0
        w[3, 7] = -3.0
        out = np.empty_like(w)
        assert lib.qv_debug_int8_roundtrip(w.ctypes.data, N, K, out.ctypes.data) == 0
        ref = quant_dequant_int8(w)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (N, K)
        assert np.abs(out - w).max() <= np.abs(w).max(axis=1).max() / 127 * 0.51 + 1e-7
    assert lib.qv_debug_int8_roundtrip(w.ctypes.data, 60, 64, out.ctypes.data) != 0


def test_engine_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from offline_tarteel_amd.engine import Engine, QvError

    with pytest.raises(QvError):
        Engine(with_model=False)


def test_product_never_imports_oracle():
    from pathlib import Path

    import re

    pkg = Path(__file__).resolve().parent.parent / "offline-tarteel_amd"
    py_import = re.compile(r"^\s*(from|import)\s+oracle\b|importlib[^\n]*oracle|libqv_oracle", re.M)
    c_use = re.compile(r"#include[^\n]*oracle|libqv_oracle|qvo_[a-z_]+\s*\(")
    for p in pkg.rglob("*.py"):
        assert not py_import.search(p.read_text(encoding="utf-8", errors="ignore")), p
    for p in list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        assert not c_use.search(p.read_text(encoding="utf-8", errors="ignore")), p


def test_weight_spec_and_seeded_init_match_the_oracle_mirror():
    """Host-only entry points (no GPU): the tensors a weight file must hold, and the engine's seeded
    synthetic initialisation, against oracle/fastconformer_ref.py (what the GPU parity tests feed the
    fp32 reference with).  tools/convert_weights.py is written against these entry points."""
    import importlib.util
    from pathlib import Path

    import numpy as np

    from oracle import fastconformer_ref as R

    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("convert_weights", str(root / "tools" / "convert_weights.py"))
    cw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cw)
    lib = cw._lib()
    shapes = cw.weight_shapes(lib)
    ref = R.weight_shapes()
    assert list(shapes) == list(ref)
    assert all(tuple(shapes[k]) == tuple(ref[k]) for k in ref)
    mine, theirs = cw.random_weights(lib, shapes, 20260630), R.random_weights(20260630)
    for k in list(shapes)[:40] + list(shapes)[-8:]:
        assert np.array_equal(mine[k], theirs[k].numpy()), k


def test_no_product_kernel_spills_or_uses_scratch(tmp_path):
    """Every gfx950 kernel in libqverse.so: no VGPR / SGPR spills, no private (scratch) segment, at most 256 VGPRs.
    The GEMM loaders issue `buffer_load_dwordx4` from inline asm and release the data with hand-counted `s_waitcnt
    vmcnt(N)`: a spill or a scratch access inside those loops would add vector-memory operations the counts do not know
    about.  Read from the code objects' AMDGPU metadata (llvm-objdump --offloading, llvm-readelf --notes)."""
    import re
    import shutil
    import subprocess
    from pathlib import Path

    llvm = Path("/opt/rocm/lib/llvm/bin")
    if not (llvm / "llvm-objdump").exists() or not (llvm / "llvm-readelf").exists():
        import pytest

        pytest.skip("ROCm llvm tools not present")
    lib = Path(__file__).resolve().parent.parent / "offline-tarteel_amd" / "libqverse.so"
    shutil.copy(lib, tmp_path / "libqverse.so")
    subprocess.run([str(llvm / "llvm-objdump"), "--offloading", "libqverse.so"], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(tmp_path.glob("libqverse.so.*gfx950"))
    assert objs, "no gfx950 code object found in libqverse.so"
    kernels = {}
    for o in objs:
        notes = subprocess.run([str(llvm / "llvm-readelf"), "--notes", str(o)], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            kernels[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                             for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    assert len(kernels) >= 60 and any("k_gemm256" in k for k in kernels) and any("k_attention_ws" in k for k in kernels)
    # no kernel spills vector registers or needs more than 256 of them
    bad = {k: v for k, v in kernels.items() if v["vgpr_spill_count"] or v["vgpr_count"] > 256}
    assert not bad, bad
    # the hand-scheduled families (inline-asm loads + counted waits, or counted direct-to-LDS loads): no scratch at all
    counted = {k: v for k, v in kernels.items() if "k_gemm" in k or "k_attention" in k}
    assert len(counted) >= 30
    bad = {k: v for k, v in counted.items() if v["private_segment_fixed_size"] or v["sgpr_spill_count"]}
    assert not bad, bad
    # elsewhere a private segment only where a kernel indexes a small local array dynamically (the long-utterance CTC
    # variants), and SGPR spills (lane writes, no memory) only in the integer text kernels
    scratch = {k for k, v in kernels.items() if v["private_segment_fixed_size"]}
    assert all("k_ctc" in k for k in scratch), scratch
    sgpr = {k: v["sgpr_spill_count"] for k, v in kernels.items() if v["sgpr_spill_count"]}
    assert all(any(t in k for t in ("k_lcs_full", "k_frag", "k_spans", "k_track")) for k in sgpr), sgpr


def test_library_has_no_packed_fp32_instructions(tmp_path):
    """No v_pk_add/mul/fma_f32 in any product kernel but the two k_gemm_pk instantiations: round 5 traced the cross-kernel disturbance of k_logmel (DESIGN.md
    section 4, tests/test_gpu_interference.py) to packed-FP32 instructions of the victim wave delivering wrong results in
    lanes 48-63 while a wave of another kernel runs f16 MFMAs fed from LDS on the same SIMD; offline-tarteel_amd/build.py
    switches them off in the code generator (scalar f32 operations give the same bits)."""
    import re
    import shutil
    import subprocess
    from pathlib import Path

    llvm = Path("/opt/rocm/lib/llvm/bin")
    if not (llvm / "llvm-objdump").exists():
        import pytest

        pytest.skip("ROCm llvm tools not present")
    lib = Path(__file__).resolve().parent.parent / "offline-tarteel_amd" / "libqverse.so"
    shutil.copy(lib, tmp_path / "libqverse.so")
    subprocess.run([str(llvm / "llvm-objdump"), "--offloading", "libqverse.so"], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(tmp_path.glob("libqverse.so.*gfx950"))
    assert objs, "no gfx950 code object found in libqverse.so"
    n_insn, with_pk = 0, []
    for o in objs:
        dis = subprocess.run([str(llvm / "llvm-objdump"), "-d", "--no-show-raw-insn", str(o)], capture_output=True, text=True, check=True).stdout
        n_insn += dis.count("v_mfma_") + dis.count("v_fma_f32")
        kernel = None
        for ln in dis.splitlines():
            m = re.match(r"[0-9a-f]+ <(\S+)>:", ln)
            if m:
                kernel = m.group(1)
            elif re.search(r"\bv_pk_(add|mul|fma)_f32\b", ln) and kernel not in with_pk:
                with_pk.append(kernel)
    assert n_insn > 1000, "disassembly looks empty"
    # Round 6: the GEMM translation units are built without the packed forms as well.  The documented exception is exactly two
    # instantiations of the 128-wide kernel that would spill without them and carry the feature as their own symbol, k_gemm_pk
    # (csrc/qv_gemm.hip: W8A16, register-staged loaders, two stages, GLU and residual epilogue)
    bad = [k for k in with_pk if "k_gemm_pk" not in k]
    assert not bad, bad[:8]
    assert len(with_pk) <= 2, with_pk


def test_integration_md_stub_matches_the_abi():
    """INTEGRATION.md section 1 is the binding a maintainer would copy: its qv_config / qv_result structures must be
    field for field what offline-tarteel_amd/engine.py binds (which the GPU tests exercise against the library), and
    every entry point the document names must exist in include/qverse.h (the stub had drifted)."""
    import ctypes as C
    import re
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    text = (root / "INTEGRATION.md").read_text(encoding="utf-8")
    block = text.split("```python", 1)[1].split("```", 1)[0]
    src = block.split("cfg = QvConfig()")[0]                       # the two structure definitions only
    src = src.replace('lib = C.CDLL("libqverse.so")', "lib = None")
    ns = {}
    exec(compile(src.replace("import ctypes as C, numpy as np, torch", "import ctypes as C"), "INTEGRATION.md", "exec"), ns)
    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd import engine as E

    for doc_cls, eng_cls in ((ns["QvConfig"], E.QvConfig), (ns["QvResult"], E.QvResult)):
        assert [(n, t) for n, t in doc_cls._fields_] == [(n, t) for n, t in eng_cls._fields_], doc_cls.__name__
        assert C.sizeof(doc_cls) == C.sizeof(eng_cls)
    header = (root / "include" / "qverse.h").read_text(encoding="utf-8")
    named = set(re.findall(r"\b(qv_[a-z0-9_]+)\s*\(", text)) | set(re.findall(r"`(qv_[a-z0-9_]+)`", text))
    named -= {"qv_config", "qv_result"}
    missing = sorted(n for n in named if not re.search(rf"\b{n}\s*\(", header))
    assert not missing, missing


def test_product_library_carries_no_dev_hooks():
    """QVERSE_SKIP / QVERSE_DUP (drop or duplicate kernel classes, results meaningless) exist only in a
    `build.py --dev-hooks` build; the library the tests and the benchmark load must not know those names."""
    from pathlib import Path

    blob = (Path(__file__).resolve().parent.parent / "offline-tarteel_amd" / "libqverse.so").read_bytes()
    assert b"QVERSE_SKIP" not in blob and b"QVERSE_DUP" not in blob
