"""The committed static tables (offline-tarteel_amd/data/qverse_tables.bin) are exactly what
tools/build_tables.py derives from the reference's DATA files (data/quran.json and the
SentencePiece model).  Runs only where the reference checkout is mounted (the build container)."""

import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
QURAN = Path("/root/reference/data/quran.json")
TOKENIZER = Path("/root/reference/web/frontend/public/tokenizer.model")


@pytest.mark.skipif(not (QURAN.exists() and TOKENIZER.exists()), reason="reference data files not mounted")
def test_tables_rebuild_bit_identical(tmp_path):
    out = tmp_path / "tables.bin"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "build_tables.py"), "--quran", str(QURAN),
                        "--tokenizer", str(TOKENIZER), "--out", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    committed = (ROOT / "offline-tarteel_amd" / "data" / "qverse_tables.bin").read_bytes()
    assert out.read_bytes() == committed
