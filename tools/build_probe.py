"""Build tools/interference_probe_{current,withdrawn} (hipcc, gfx950) -- the regression probe behind
tests/test_gpu_interference.py: does a kernel on one stream change what the log-mel kernel computes on another?

    python tools/build_probe.py [--force]

`current` links the product's own translation units (qv_layers.hip, qv_ort.hip) as they are in the tree, `withdrawn`
replaces qv_ort.hip by tools/withdrawn/qv_ort_conv0_mfma.hip, the round-4 kernel that was seen to disturb k_logmel.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SRC = ROOT / "tools" / "interference_probe.hip"
DEPS = [Path(__file__).resolve(), ROOT / "offline-tarteel_amd" / "build.py", SRC, ROOT / "tools" / "logmel_variants.h", ROOT / "tools" / "withdrawn" / "qv_ort_conv0_mfma.hip",
        *sorted((ROOT / "offline-tarteel_amd" / "csrc").glob("*.h")), *sorted((ROOT / "offline-tarteel_amd" / "csrc").glob("*.inc")),
        ROOT / "offline-tarteel_amd" / "csrc" / "qv_gemm.hip", ROOT / "offline-tarteel_amd" / "csrc" / "qv_gemm256.hip",
        ROOT / "offline-tarteel_amd" / "csrc" / "qv_layers.hip", ROOT / "offline-tarteel_amd" / "csrc" / "qv_ort.hip"]


def build(force: bool = False) -> list[Path]:
    from concurrent.futures import ThreadPoolExecutor

    outs, jobs = [], []
    newest = max(d.stat().st_mtime for d in DEPS)
    # current / withdrawn are compiled like the product's non-GEMM translation units (offline-tarteel_amd/build.py: no packed-FP32 instructions);
    # withdrawn_pk is the positive control: the same translation unit with v_pk_add/mul/fma_f32 left on, i.e. the victims as
    # they were compiled until round 5 (the withdrawn aggressor's range pass has no such instruction, so only the victims
    # change between withdrawn and withdrawn_pk)
    for flavour in ("current", "withdrawn", "withdrawn_pk"):
        out = ROOT / "tools" / f"interference_probe_{flavour}"
        outs.append(out)
        if not force and out.exists() and out.stat().st_mtime >= newest:
            continue
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", str(ROOT / "offline-tarteel_amd" / "csrc"),
               "-I", str(ROOT / "include"), *(["-DPROBE_WITHDRAWN"] if flavour.startswith("withdrawn") else []),
               *([] if flavour.endswith("_pk") else ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]), str(SRC), "-o", str(out)]
        jobs.append((out, cmd))

    def cc(job):
        out, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {out.name}:\n{r.stderr[-4000:]}")

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(cc, jobs))
    return outs


if __name__ == "__main__":
    for o in build("--force" in sys.argv):
        print(o)
