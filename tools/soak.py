#!/usr/bin/env python
"""dev: randomised soak of the engine with batches in flight -- ragged batches of random size and clip length (0.05-30 s)
through predict_batch_async on a four-context engine, every batch compared with the same batch through a one-context
engine (bit-for-bit the same rows).  Looks for races between contexts, staging-slot reuse and shape-dependent paths.

    python tools/soak.py [--batches 300] [--seed 1] [--precision 0]
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--precision", type=int, default=0)
    ap.add_argument("--contexts", type=int, default=4)
    ap.add_argument("--recipes", type=int, default=0,
                    help="draw every batch from this many fixed (batch size, length multiset) recipes, each with its own staging buffer, "
                         "lengths permuted and audio rotated per batch: the forward-graph replay path (keys repeat, more recipes than graph slots evict)")
    ap.add_argument("--third", action="store_true", help="a second one-context engine says which side of a mismatch is the odd one")
    args = ap.parse_args()
    import numpy as np
    import torch

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from synth import synth_audio

    cap = 480000
    rng = np.random.default_rng(args.seed)
    pool = torch.from_numpy(synth_audio(64, cap, seed=args.seed)).cuda().contiguous()
    e4 = Engine(device=0, with_model=True, seed=5, precision=args.precision, max_batch=64, max_samples=cap, contexts=args.contexts)
    e1 = Engine(device=0, with_model=True, seed=5, precision=args.precision, max_batch=64, max_samples=cap, contexts=1)
    e1b = Engine(device=0, with_model=True, seed=5, precision=args.precision, max_batch=64, max_samples=cap, contexts=1) if args.third else None
    key = lambda r: (r["surah"], r["ayah"], r["ayah_end"], r["source"], r["score"], r["t_frames"], r["n_candidates"])  # noqa: E731
    inflight, bad, t0, utts = [], 0, time.perf_counter(), 0
    recipes = []
    for _ in range(args.recipes):
        B = int(rng.integers(2, 65))
        hi = 160000 if rng.random() < 0.7 else cap
        lens = [int(x) for x in rng.integers(800, hi + 1, size=B)]
        recipes.append({"lens": lens, "buf": torch.zeros(B, max(lens), device="cuda")})
    try:
        for i in range(args.batches):
            if recipes:
                busy = {id(x[1]) for x in inflight}
                free = [r for r in recipes if id(r["buf"]) not in busy]      # a staging buffer is rewritten only once its batch has been joined
                r = free[int(rng.integers(0, len(free)))]
                lens = [r["lens"][j] for j in rng.permutation(len(r["lens"]))]
                B, n, a = len(lens), max(lens), r["buf"]
                a.copy_(pool.roll(int(rng.integers(0, 64)), 0)[:B, :n])
            else:
                B = int(rng.integers(1, 65))
                kind = rng.random()
                hi = 16000 if kind < 0.2 else 160000 if kind < 0.8 else cap
                lens = [int(x) for x in rng.integers(800, hi + 1, size=B)]
                n = max(lens)
                a = pool[:B, :n].contiguous()
            for b, L in enumerate(lens):
                a[b, L:] = 0
            ctx = e4.predict_batch_async(a, lens)
            inflight.append((ctx, a, lens))
            utts += B
            if len(inflight) == e4.contexts:
                c, aa, ll = inflight.pop(0)
                got = [key(r) for r in e4.fetch_results(c, len(ll), e4.frames_for(max(ll)), want_text=False)]
                want = [key(r) for r in e1.predict_batch(aa, ll, want_text=False)]
                if got != want:
                    bad += 1
                    diff = [j for j, (g, w) in enumerate(zip(got, want)) if g != w]
                    j0 = diff[0]
                    who = ""
                    if e1b is not None:
                        third = [key(r) for r in e1b.predict_batch(aa, ll, want_text=False)]
                        who = "third engine agrees with: " + ("one-context" if third == want else "multi-context" if third == got else "neither")
                    print("MISMATCH at batch", i, "B", len(ll), "max_len", max(ll), "n_diff", len(diff), diff[:5],
                          "len", ll[j0], "got", got[j0], "want", want[j0], who, flush=True)
        while inflight:
            c, aa, ll = inflight.pop(0)
            got = [key(r) for r in e4.fetch_results(c, len(ll), e4.frames_for(max(ll)), want_text=False)]
            want = [key(r) for r in e1.predict_batch(aa, ll, want_text=False)]
            bad += got != want
    finally:
        graph = e4.forward_graph_stats()
        e4.close(); e1.close()
        if e1b is not None:
            e1b.close()
    print(f"soak: {args.batches} ragged batches, {utts} utterances, {bad} mismatching batches, {time.perf_counter() - t0:.1f} s"
          + (f"; {args.recipes} recipes" if args.recipes else "") + f"; forward graph of the multi-context engine: {graph}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
