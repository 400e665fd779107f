"""Compile the C restatement (oracle/qv_oracle.c) into oracle/libqv_oracle.so.

Test infrastructure: called by __graft_entry__.build() and by tests; the product never
loads the result.  -ffp-contract=off keeps the blended fragment score bit-exact.
"""

from __future__ import annotations

import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
SRC = HERE / "qv_oracle.c"
LIB = HERE / "libqv_oracle.so"


def build(force: bool = False) -> Path:
    if not force and LIB.exists() and LIB.stat().st_mtime >= SRC.stat().st_mtime:
        return LIB
    cmd = ["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
           "-o", str(LIB), str(SRC), "-lm"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
