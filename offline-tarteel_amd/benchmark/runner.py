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
Several GPUs (not in the reference): launched as

    python -m torch.distributed.run --nproc-per-node N -m offline_tarteel_amd.benchmark.runner --experiment ... --batch 16

one process per GPU: rank 0 reads the manifest and the clip lengths, ``dist.shard_plan`` deals the length-sorted files,
every rank runs the plugin on its own share (its engine lives on cuda:LOCAL_RANK), ``dist.all_gather_results`` -- the
path's only exchange, 16 B per file -- restores the manifest order, and rank 0 scores and writes the result files,
which are identical to a single-process run's apart from the latencies (``run_experiment_sharded``).
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
# QVERSE_EXPERIMENTS_DIR: a second directory of <name>/run.py plugins (the reference discovers experiments by scanning
# its experiments/ directory, runner.py:146-208; the tests point this at a stub plugin that needs no GPU)
_extra = os.getenv("QVERSE_EXPERIMENTS_DIR")
if _extra and Path(_extra).is_dir():
    for _d in sorted(Path(_extra).iterdir()):
        if (_d / "run.py").exists():
            EXPERIMENT_REGISTRY.setdefault(_d.name, _d / "run.py")


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


def _predict_group(mod, paths: list[str], use_batch: bool):
    """one engine call for a group of files; on failure the group is retried file by file, as the reference isolates
    failures per sample (runner.py:297-325).  Returns (results or None per file, seconds per file, error strings)."""
    try:
        t0 = time.perf_counter()
        results = mod.predict_batch(paths) if use_batch else [mod.predict(paths[0])]
        dt = (time.perf_counter() - t0) / len(paths)
        return list(results), [dt] * len(paths), [None] * len(paths)
    except Exception as e:
        print(f"  Error on {[Path(p).name for p in paths]}: {e}")
        if len(paths) == 1:
            return [None], [0.0], [f"{type(e).__name__}: {e}"]
    results, secs, errors = [], [], []
    for path in paths:
        try:
            t0 = time.perf_counter()
            results.append(mod.predict(path))
            secs.append(time.perf_counter() - t0)
            errors.append(None)
        except Exception as e1:
            print(f"  Error on {Path(path).name}: {e1}")
            results.append(None)
            secs.append(0.0)
            errors.append(f"{type(e1).__name__}: {e1}")
    return results, secs, errors


def run_experiment_sharded(exp: dict, samples: list[dict] | None, corpus_dir: Path, batch: int = 1, deal: str = "strided",
                           group=None) -> dict | None:
    """run_experiment for predict()-style experiments over the ranks of a torch.distributed job (one process per GPU).

    rank 0 passes the manifest rows (other ranks pass None): it drops the rows whose audio is absent, reads the clip
    lengths from the file headers, and broadcasts rows + shard plan.  Every rank predicts its share in engine calls of
    ``batch`` files (longest first), packs (surah, ayah, ayah_end, score) into int32[per_rank, 4] and all-gathers; rank
    0 scores in manifest order and returns the result dict of run_experiment -- every other rank returns None."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from .. import dist as qdist
    from ..audio import probe_samples

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mod = load_module(exp["name"].replace("/", "_").replace("-", "_"), exp["run_path"])
    if not hasattr(mod, "predict"):
        raise SystemExit(f"{exp['name']}: the sharded runner needs predict(); transcribe()-only experiments run single-process")
    box = [None]
    if rank == 0:
        present = [s for s in samples if (corpus_dir / s["file"]).exists()]
        lengths = [probe_samples(str(corpus_dir / s["file"])) for s in present]
        order, _ = qdist.shard_plan(lengths, world, deal)
        box = [(present, order.tolist())]
    dist.broadcast_object_list(box, src=0, group=group)
    present, order = box[0]
    if not present:      # nothing to predict (no audio under the corpus directory): no collective with empty rows
        return None if rank != 0 else {"name": exp["name"], "recall": 0, "precision": 0, "sequence_accuracy": 0, "total": 0,
                                       "avg_latency": 0, "model_size": 0, "per_sample": [], "world_size": world}
    order = np.asarray(order, dtype=np.int64)
    per = len(order) // world
    mine = [int(i) for i in order[rank * per: (rank + 1) * per]]
    todo = [i for i in mine if i >= 0]
    if todo:   # warm-up on this rank's first file; failure is only reported (runner.py:272-280)
        try:
            mod.predict(str(corpus_dir / present[todo[0]]["file"]))
        except Exception as e:
            print(f"  [rank {rank}] Warmup failed for {exp['name']}: {e}")
    use_batch = batch > 1 and hasattr(mod, "predict_batch")
    got: dict[int, tuple] = {}
    for s0 in range(0, len(todo), batch if use_batch else 1):
        idx = todo[s0: s0 + (batch if use_batch else 1)]
        res, secs, errs = _predict_group(mod, [str(corpus_dir / present[i]["file"]) for i in idx], use_batch)
        for i, r, t, e in zip(idx, res, secs, errs):
            got[i] = (r, t, e)
    rows = qdist.pack_results([got[i][0] if i >= 0 else None for i in mine])
    side = np.zeros((per, 2), dtype=np.float32)                      # harness bookkeeping: seconds per file, error flag
    for k, i in enumerate(mine):
        if i >= 0:
            side[k] = (got[i][1], 1.0 if got[i][2] else 0.0)
    dev = torch.device(f"cuda:{torch.cuda.current_device()}") if dist.get_backend(group) == "nccl" else torch.device("cpu")
    full = qdist.all_gather_results(torch.from_numpy(rows).to(dev), order, len(present), group=group)
    side_full = qdist.all_gather_rows(torch.from_numpy(side).to(dev), order, len(present), group=group)
    if rank != 0:
        return None
    try:
        size = mod.model_size()
    except Exception:
        size = 0
    per_sample, tot = [], {"recall": 0.0, "precision": 0.0, "sequence_accuracy": 0.0}
    for sample, res, (secs, failed) in zip(present, qdist.unpack_results(full, round_dp=4), side_full.tolist()):
        expected = sample.get("expected_verses", [{"surah": sample["surah"], "ayah": sample["ayah"]}])
        em = predict_to_emissions(res)
        sc = score_sequence(expected, em)
        for k in tot:
            tot[k] += sc[k]
        row = {"id": sample["id"], "expected": expected, "predicted": em, **sc, "latency": float(secs)}
        if failed:
            row["error"] = "predict failed on its rank (see that rank's log)"
        per_sample.append(row)
    n = len(per_sample)
    return {
        "name": exp["name"], "recall": tot["recall"] / n if n else 0, "precision": tot["precision"] / n if n else 0,
        "sequence_accuracy": tot["sequence_accuracy"] / n if n else 0, "total": n,
        "avg_latency": sum(r["latency"] for r in per_sample) / n if n else 0, "model_size": size, "per_sample": per_sample,
        "world_size": world,
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
    ap.add_argument("--deal", type=str, default="strided", choices=["strided", "contiguous"],
                    help="several GPUs: how dist.shard_plan deals the length-sorted files to the ranks")
    args = ap.parse_args(argv)
    corpus = Path(args.corpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        return main_sharded(args, corpus)
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


def main_sharded(args, corpus: Path):
    """one process per GPU under torch.distributed.run: see the module docstring"""
    import torch.distributed as dist

    from .. import dist as qdist

    if args.mode != "full":
        raise SystemExit("--mode streaming runs single-process (the verse tracker is sequential per file)")
    rank, world, _ = qdist.init_process_group()
    samples = None
    if rank == 0:
        samples = load_manifest(corpus)
        if args.category:
            samples = [s for s in samples if s["category"] == args.category]
            print(f"Filtered to {len(samples)} samples in category '{args.category}'")
    experiments = discover_experiments(args.experiment)
    if not experiments:
        if rank == 0:
            print(f"No experiments found matching '{args.experiment}'")
        dist.destroy_process_group()
        return []
    if rank == 0:
        print(f"Running {len(experiments)} experiment(s) on {len(samples)} sample(s) [full transcript, {world} ranks, {args.deal} deal]...")
    results = []
    for exp in experiments:
        if rank == 0:
            print(f"\n>>> {exp['name']}")
        r = run_experiment_sharded(exp, samples, corpus, batch=args.batch, deal=args.deal)
        if r is not None:
            results.append(r)
            print(f"    Recall: {r['recall']:.0%}  Precision: {r['precision']:.0%}  SeqAcc: {r['sequence_accuracy']:.0%}")
    if rank == 0:
        print_table(results)
        save_results(results, mode=args.mode, category=args.category, chunk_seconds=args.chunk)
    dist.barrier()
    dist.destroy_process_group()
    return results


if __name__ == "__main__":
    sys.path.insert(0, str(PKG.parent))
    main()
