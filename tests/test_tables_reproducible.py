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


@pytest.mark.skipif(not (QURAN.exists() and TOKENIZER.exists()), reason="reference data files not mounted")
def test_emit_json_matches_the_upstream_artefact_shape(tmp_path):
    """--emit-json writes quran_ctc_tokens.json as PLAN.md:102-103 describes the upstream file: 35,717 keys
    "surah:ayah:ayah_end", 29,481 of them multi-ayah spans -- which is every span of at most SIX ayat inside
    a surah (five would give 30,043 keys; the prose at PLAN.md:121-124 says five, the counts say six) -- ids of
    the text tokenised as joined text; single-verse entries equal the binary table's."""
    import json

    import numpy as np

    import offline_tarteel_amd
    from offline_tarteel_amd.tables import Tables

    out, js = tmp_path / "tables.bin", tmp_path / "quran_ctc_tokens.json"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "build_tables.py"), "--quran", str(QURAN), "--tokenizer",
                        str(TOKENIZER), "--out", str(out), "--emit-json", str(js)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    table = json.loads(js.read_text())
    assert len(table) == 35717
    assert sum(1 for k in table if k.split(":")[1] != k.split(":")[2]) == 29481
    tb = Tables(offline_tarteel_amd.TABLES_PATH)
    for s, a in ((1, 1), (2, 255), (112, 2), (114, 6)):
        assert table[f"{s}:{a}:{a}"] == tb.token_ids(tb.verse_index(s, a), 1).tolist()
    # a span that does not start at a bismillah verse is the same in both span-text conventions
    assert table["2:2:4"] == tb.token_ids(tb.verse_index(2, 2), 3).tolist()
    assert max(int(k.split(":")[2]) - int(k.split(":")[1]) for k in table) == 5
