"""End-to-end golden table of the v1 corpus, as data.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/gen_v1_expected.py

Joins the reference's manifest (benchmark/test_corpus/manifest.json: id, file, category, expected verses)
with its published per-sample outputs -- benchmark/results/2026-06-28_135450.json (c2c-direct-mixed) and
2026-06-28_135603.json (c2c-direct-mixed-tta) -- into tests/golden/v1_expected.json.  These are the only
pins the reference holds for the acoustic model: tools/v1_parity.py compares a run of this
repo's runner against them the day the weight file (QVERSE_WEIGHTS) and the audio files are available.
"""
import json
from pathlib import Path

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent


def per_sample(path, name):
    doc = json.loads(path.read_text())
    (entry,) = [e for e in doc if e["name"] == name]
    return entry, {s["id"]: s for s in entry["per_sample"]}


def main():
    manifest = json.loads((REF / "benchmark/test_corpus/manifest.json").read_text(encoding="utf-8"))["samples"]
    mixed, mixed_rows = per_sample(REF / "benchmark/results/2026-06-28_135450.json", "c2c-direct-mixed")
    tta, tta_rows = per_sample(REF / "benchmark/results/2026-06-28_135603.json", "c2c-direct-mixed-tta")
    rows = []
    for s in manifest:
        m, t = mixed_rows[s["id"]], tta_rows[s["id"]]
        rows.append({
            "id": s["id"], "file": s["file"], "category": s["category"],
            "file_in_reference_tree": (REF / "benchmark/test_corpus" / s["file"]).exists(),
            "expected": s.get("expected_verses", [{"surah": s["surah"], "ayah": s["ayah"]}]),
            "mixed": {"predicted": m["predicted"], "recall": m["recall"], "sequence_accuracy": m["sequence_accuracy"]},
            "tta": {"predicted": t["predicted"], "recall": t["recall"], "sequence_accuracy": t["sequence_accuracy"]},
        })
    doc = {
        "source": {"manifest": "benchmark/test_corpus/manifest.json",
                   "c2c-direct-mixed": "benchmark/results/2026-06-28_135450.json",
                   "c2c-direct-mixed-tta": "benchmark/results/2026-06-28_135603.json"},
        "summary": {k: {x: e[x] for x in ("recall", "precision", "sequence_accuracy", "total", "avg_latency", "model_size")}
                    for k, e in (("c2c-direct-mixed", mixed), ("c2c-direct-mixed-tta", tta))},
        "samples": rows,
    }
    out = HERE / "v1_expected.json"
    out.write_text(json.dumps(doc, ensure_ascii=False, indent=1) + "\n", encoding="utf-8")
    print(f"wrote {out} ({len(rows)} samples, {sum(r['file_in_reference_tree'] for r in rows)} audio files present in the reference tree)")


if __name__ == "__main__":
    main()
