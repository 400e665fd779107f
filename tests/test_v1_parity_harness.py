"""The v1-corpus golden table (reference result files) and the harness that
compares a runner result with it -- exercised on CPU with the reference's own rows standing in for
a run (the real run needs the weight file; tools/v1_parity.py then skips cleanly)."""

import copy
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


def test_golden_table_is_the_published_one(golden_dir):
    doc = json.loads((golden_dir / "v1_expected.json").read_text(encoding="utf-8"))
    assert len(doc["samples"]) == 53
    assert doc["summary"]["c2c-direct-mixed"]["recall"] == 52 / 53 and doc["summary"]["c2c-direct-mixed-tta"]["recall"] == 1.0
    by = {s["id"]: s for s in doc["samples"]}
    # the one published miss of the mixed plugin, fixed by TTA (EXPERIMENTS.md / SURVEY.md section 6)
    assert by["retasy_004"]["mixed"]["predicted"] == [{"surah": 56, "ayah": 36, "score": 0.0}]
    assert [(p["surah"], p["ayah"]) for p in by["retasy_004"]["tta"]["predicted"]] == [(114, 3)]
    assert [(p["surah"], p["ayah"]) for p in by["multi_114_001_006"]["mixed"]["predicted"]] == [(114, a) for a in range(1, 7)]
    assert sum(s["file_in_reference_tree"] for s in doc["samples"]) == 44


def test_compare_rows_flags_differences_and_skips(golden_dir):
    import v1_parity

    doc = json.loads((golden_dir / "v1_expected.json").read_text(encoding="utf-8"))
    run = {"recall": 52 / 53, "per_sample": [
        {"id": s["id"], "predicted": copy.deepcopy(s["mixed"]["predicted"]), "latency": 0.1} for s in doc["samples"]]}
    rep = v1_parity.compare_rows(doc, "c2c-direct-mixed", run, 1e-2)
    assert rep["compared"] == 53 and rep["differing"] == 0 and rep["skipped"] == 0
    run["per_sample"][3]["predicted"][0]["score"] += 0.02            # beyond the slack
    run["per_sample"][5]["predicted"][0]["ayah"] += 1                 # another verse
    # the runner's error convention (empty prediction, latency 0.0): only an undecodable file is excused ...
    run["per_sample"][7] = {"id": run["per_sample"][7]["id"], "predicted": [], "latency": 0.0,
                            "error": "ValueError: x.mp3: not a RIFF/WAVE file (no decoder for compressed audio in this environment)"}
    # ... an engine / capacity error is a row the reference handled and this path did not
    run["per_sample"][8] = {"id": run["per_sample"][8]["id"], "predicted": [], "latency": 0.0,
                            "error": "QvError: qv_predict_batch failed (4): audio longer than engine capacity"}
    run["per_sample"][11] = {"id": run["per_sample"][11]["id"], "predicted": [], "latency": 0.0}   # (no error text: same)
    del run["per_sample"][9]                                          # file absent from the corpus directory
    rep = v1_parity.compare_rows(doc, "c2c-direct-mixed", run, 1e-2)
    assert rep["differing"] == 4 and rep["skipped"] == 2 and rep["compared"] == 51
    by = {r["id"]: r for r in rep["rows"]}
    assert by[doc["samples"][8]["id"]]["status"] == "DIFFERS" and "capacity" in by[doc["samples"][8]["id"]]["error"]
    assert by[doc["samples"][7]["id"]]["status"].startswith("skipped (undecodable audio")


def test_harness_skips_cleanly_without_weights():
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "v1_parity.py")], capture_output=True, text=True,
                       env={"PATH": "/usr/bin:/bin"}, timeout=120)
    assert p.returncode == 77 and "SKIPPED" in p.stdout
