"""Benchmark harness with the reference's CLI and result schema (benchmark/runner.py):

    python -m offline_tarteel_amd.benchmark.runner --experiment c2c-direct-mixed \
        [--category short] [--corpus /path/to/benchmark/test_corpus] [--batch 16]

* experiments are loaded BY FILE PATH from experiments/<name>/run.py (hyphenated dirs) and must
  export predict(audio_path) -> dict and model_size() -> int            (runner.py:89-94,248-289)
* a manifest row is skipped when its audio file is absent; any exception inside predict counts
  the sample as an empty prediction with latency 0.0                    (runner.py:297-325)
* recall / precision / sequence accuracy by ordered subsequence match  (runner.py:104-143)
* results/<timestamp>.json (full per-sample) + latest.json (best per key) (runner.py:386-469)

``--batch N`` (not in the reference) sends N files per engine call through the plugin's
predict_batch; per-sample latency is then the call time divided by N.
``--mode`` / ``--chunk`` follow the reference (runner.py:250,309-321,350-351,419): an experiment
that exports predict() is run through it in both modes (streaming only changes the result label
and records chunk_seconds); a transcribe()-only experiment goes through StreamingPipeline --
run_on_audio_chunked in streaming mode, run_on_full_transcript otherwise -- with the MI355X
verse tracker / match_verse underneath (offline-tarteel_amd/streaming.py).
"""

from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import time
from datetime import datetime
from pathlib import Path

PKG = Path(__file__).resolve().parent.parent
EXPERIMENTS_DIR = PKG / "experiments"
RESULTS_DIR = Path(os.getenv("QVERSE_RESULTS_DIR", str(Path(__file__).resolve().parent / "results")))
DEFAULT_CORPUS = Path(os.getenv("QVERSE_CORPUS_DIR", str(Path(__file__).resolve().parent / "test_corpus")))

EXPERIMENT_REGISTRY = {
    "c2c-direct-mixed": EXPERIMENTS_DIR / "c2c-direct-mixed" / "run.py",
    "c2c-direct-mixed-tta": EXPERIMENTS_DIR / "c2c-direct-mixed-tta" / "run.py",
}


def load_module(name: str, file_path: Path):
    spec = importlib.util.spec_from_file_location(name, str(file_path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_manifest(corpus_dir: Path) -> list[dict]:
    with open(corpus_dir / "manifest.json", encoding="utf-8") as f:
        return json.load(f)["samples"]


def score_sequence(expected: list[dict], predicted: list[dict]) -> dict:
    """expected verses must appear in the prediction in order; precision counts the predicted
    entries that were used; sequence accuracy is exact list equality."""
    if not expected:
        return {"recall": 1.0, "precision": 1.0, "sequence_accuracy": 1.0}
    if not predicted:
        return {"recall": 0.0, "precision": 0.0, "sequence_accuracy": 0.0}
    want = [(e["surah"], e["ayah"]) for e in expected]
    got = [(p["surah"], p["ayah"]) for p in predicted]
    cursor, used = 0, set()
    for w in want:
        try:
            j = got.index(w, cursor)
        except ValueError:
            continue
        used.add(j)
        cursor = j + 1
    return {"recall": len(used) / len(want), "precision": len(used) / len(got),
            "sequence_accuracy": 1.0 if got == want else 0.0}


def predict_to_emissions(result: dict) -> list[dict]:
    """one emission per ayah of [ayah, ayah_end], sharing the score (runner.py:211-228)."""
    if not result or result.get("surah", 0) == 0:
        return []
    first = result["ayah"]
    last = result.get("ayah_end") or first
    score = result.get("score", 0.0)
    return [{"surah": result["surah"], "ayah": a, "score": score} for a in range(first, last + 1)]


def discover_experiments(name: str | None) -> list[dict]:
    names = [name] if name else list(EXPERIMENT_REGISTRY)
    return [{"name": n, "run_path": EXPERIMENT_REGISTRY[n], "model_name": None}
            for n in names if n in EXPERIMENT_REGISTRY]


def run_experiment(exp: dict, samples: list[dict], corpus_dir: Path, batch: int = 1, mode: str = "full",
                   chunk_seconds: float = 3.0, pipeline=None) -> dict | None:
    mod = load_module(exp["name"].replace("/", "_").replace("-", "_"), exp["run_path"])
    use_predict = hasattr(mod, "predict")
    if not use_predict and not hasattr(mod, "transcribe"):
        print(f"  Skipping {exp['name']} -- no predict() or transcribe() function")
        return None
    if not use_predict and pipeline is None:
        from ..streaming import StreamingPipeline

        pipeline = StreamingPipeline()
    present = [s for s in samples if (corpus_dir / s["file"]).exists()]
    if present:  # warm-up on the first sample; failure is only reported
        try:
            (mod.predict if use_predict else mod.transcribe)(str(corpus_dir / present[0]["file"]))
        except Exception as e:
            print(f"  Warmup failed for {exp['name']}: {e}")
    try:
        size = mod.model_size()
    except Exception:
        size = 0
    use_batch = use_predict and batch > 1 and hasattr(mod, "predict_batch")
    per_sample, latencies = [], []
    tot = {"recall": 0.0, "precision": 0.0, "sequence_accuracy": 0.0}
    for s0 in range(0, len(present), batch if use_batch else 1):
        group = present[s0: s0 + (batch if use_batch else 1)]
        paths = [str(corpus_dir / s["file"]) for s in group]
        errors = [None] * len(group)
        try:
            t0 = time.perf_counter()
            if use_predict:
                results = mod.predict_batch(paths) if use_batch else [mod.predict(paths[0])]
                emissions = [predict_to_emissions(r) for r in results]
            elif mode == "streaming":
                emissions = [pipeline.run_on_audio_chunked(paths[0], mod.transcribe, chunk_seconds=chunk_seconds)]
            else:
                emissions = [pipeline.run_on_full_transcript(paths[0], mod.transcribe)]
            elapsed = (time.perf_counter() - t0) / len(group)
        except Exception as e:
            print(f"  Error on {[s['id'] for s in group]}: {e}")
            emissions, elapsed = [[] for _ in group], 0.0
            errors = [f"{type(e).__name__}: {e}"] * len(group)
            if use_batch and len(group) > 1:
                # one undecodable / over-long file must not empty the whole group: the reference isolates
                # failures per sample (runner.py:297-325), so the group is retried file by file
                emissions, per_file, errors = [], [], []
                for path in paths:
                    try:
                        t0 = time.perf_counter()
                        emissions.append(predict_to_emissions(mod.predict(path)))
                        per_file.append(time.perf_counter() - t0)
                        errors.append(None)
                    except Exception as e1:
                        print(f"  Error on {Path(path).name}: {e1}")
                        emissions.append([])
                        per_file.append(0.0)
                        errors.append(f"{type(e1).__name__}: {e1}")
                elapsed = per_file
        if not isinstance(elapsed, list):
            elapsed = [elapsed] * len(group)
        for (sample, em), elapsed, err in zip(zip(group, emissions), elapsed, errors):
            expected = sample.get("expected_verses", [{"surah": sample["surah"], "ayah": sample["ayah"]}])
            sc = score_sequence(expected, em)
            for k in tot:
                tot[k] += sc[k]
            latencies.append(elapsed)
            row = {"id": sample["id"], "expected": expected, "predicted": em, **sc, "latency": elapsed}
            if err:   # (additive key: which exception emptied this row -- tools/v1_parity.py tells undecodable audio from engine errors)
                row["error"] = err
            per_sample.append(row)
    n = len(per_sample)
    return {
        "name": exp["name"] if mode == "full" else f"{exp['name']} (stream {chunk_seconds:.0f}s)",
        "recall": tot["recall"] / n if n else 0, "precision": tot["precision"] / n if n else 0,
        "sequence_accuracy": tot["sequence_accuracy"] / n if n else 0,
        "total": n, "avg_latency": sum(latencies) / n if n else 0, "model_size": size, "per_sample": per_sample,
    }


def print_table(results: list[dict]):
    print()
    print(f"{'Experiment':<30} {'Recall':>8} {'Precision':>10} {'SeqAcc':>8} {'Latency':>10} {'Size':>10}")
    print("-" * 78)
    for r in results:
        print(f"{r['name']:<30} {r['recall']:>8.0%} {r['precision']:>10.0%} {r['sequence_accuracy']:>8.0%} "
              f"{r['avg_latency']:>9.3f}s {r['model_size'] / (1024 ** 3):>8.1f} GB")
    print()


def save_results(results: list[dict], *, mode: str = "full", category: str | None = None,
                 results_dir: Path | None = None, chunk_seconds: float = 3.0) -> Path:
    out_dir = Path(results_dir or RESULTS_DIR)
    out_dir.mkdir(parents=True, exist_ok=True)
    stamp = datetime.now().strftime("%Y-%m-%d_%H%M%S")
    path = out_dir / f"{stamp}.json"
    path.write_text(json.dumps(results, indent=2, default=str))
    latest_path = out_dir / "latest.json"
    latest = {}
    if latest_path.exists():
        for e in json.loads(latest_path.read_text()):
            latest[(e.get("name"), e.get("mode", "full"), e.get("category"), e.get("total"), e.get("chunk_seconds"))] = e
    for r in results:
        summary = {k: r[k] for k in ("name", "recall", "precision", "sequence_accuracy", "total", "avg_latency", "model_size")}
        chunk = chunk_seconds if mode == "streaming" else None
        summary.update(timestamp=stamp, mode=mode, category=category, chunk_seconds=chunk, source_file=path.name)
        key = (summary["name"], mode, category, summary["total"], chunk)
        prev = latest.get(key)
        better = prev is None or r["sequence_accuracy"] > prev.get("sequence_accuracy", 0) or (
            r["sequence_accuracy"] == prev.get("sequence_accuracy", 0)
            and r["avg_latency"] < prev.get("avg_latency", float("inf")))
        if better:
            latest[key] = summary
    ordered = sorted(latest.values(), key=lambda x: (x.get("name", ""), x.get("mode", "full"), x.get("category") or "",
                                                      x.get("total", 0), x.get("chunk_seconds") or 0))
    latest_path.write_text(json.dumps(ordered, indent=2, default=str))
    print(f"Results saved to {path}; updated {latest_path}")
    return path


def main(argv=None):
    ap = argparse.ArgumentParser(description="Benchmark the qverse experiments (reference CLI)")
    ap.add_argument("--experiment", type=str, help="Run only this experiment")
    ap.add_argument("--category", type=str, help="Filter samples by category")
    ap.add_argument("--mode", type=str, default="full", choices=["full", "streaming"])
    ap.add_argument("--chunk", type=float, default=3.0)
    ap.add_argument("--corpus", type=str, default=str(DEFAULT_CORPUS), help="directory holding manifest.json + audio")
    ap.add_argument("--batch", type=int, default=1, help="files per engine call (uses predict_batch)")
    args = ap.parse_args(argv)
    corpus = Path(args.corpus)
    samples = load_manifest(corpus)
    if args.category:
        samples = [s for s in samples if s["category"] == args.category]
        print(f"Filtered to {len(samples)} samples in category '{args.category}'")
    experiments = discover_experiments(args.experiment)
    if not experiments:
        print(f"No experiments found matching '{args.experiment}'")
        return []
    label = "full transcript" if args.mode == "full" else f"streaming {args.chunk:g}s chunks"
    print(f"Running {len(experiments)} experiment(s) on {len(samples)} sample(s) [{label}]...")
    results = []
    for exp in experiments:
        print(f"\n>>> {exp['name']}")
        r = run_experiment(exp, samples, corpus, batch=args.batch, mode=args.mode, chunk_seconds=args.chunk)
        if r is None:
            continue
        results.append(r)
        print(f"    Recall: {r['recall']:.0%}  Precision: {r['precision']:.0%}  SeqAcc: {r['sequence_accuracy']:.0%}")
    print_table(results)
    save_results(results, mode=args.mode, category=args.category, chunk_seconds=args.chunk)
    return results


if __name__ == "__main__":
    sys.path.insert(0, str(PKG.parent))
    main()
