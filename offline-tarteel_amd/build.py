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
    flavour = "dev-hooks" if dev_hooks else "product"
    if not stamp.exists() or stamp.read_text().strip() != flavour:
        force = True
    headers = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "qverse.h"]
    jobs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC, *FLAGS, *(["-DQV_DEV_HOOKS"] if dev_hooks else []), "-c", str(src), "-o", str(obj)]
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
