#!/usr/bin/env python
"""Post-logits stages alone (greedy decode -> retrieval -> CTC rerank) on verse-shaped log-probs:
what they cost when the transcript is a real recitation rather than the near-empty string the
random-weight benchmark produces.  Log-probs come from the tests' synthetic recipe (a frame path
through a verse's token ids + hashed noise); verses are drawn with a fixed seed.

    python tools/post_bench.py [--batch 64] [--frames 126] [--steps 20]
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=126)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--case", type=int, default=-1, help="run only this case (0 = clean / gate passes ... 3 = noisy); default all")
    args = ap.parse_args()

    import numpy as np
    import torch

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from synth import synth_logits

    B, T = args.batch, args.frames
    eng = Engine(device=0, with_model=False, max_batch=B, max_samples=T * 1280 + 1280)
    rng = np.random.default_rng(20260630)
    n_verses = len(eng.tables.s["tok_off"]) // 6
    rows = []
    cases = (("clean (gate passes)", 1.0, 8.0), ("corrupted", 2.0, 6.0), ("corrupted more", 2.4, 6.0),
             ("noisy (gate fails -> CTC rerank)", 3.5, 4.0))
    for name, noise, boost in (cases if args.case < 0 else cases[args.case: args.case + 1]):
        lps, used = [], 0
        while len(lps) < B:
            v = int(rng.integers(0, n_verses))
            ids = eng.tables.token_ids(v, 1).tolist()
            if not (4 <= len(ids) and 2 * len(ids) + 1 <= T):
                continue
            lg = torch.from_numpy(synth_logits(ids, T, seed=1000 + used, noise=noise, boost=boost, rep=2))
            lps.append(torch.log_softmax(lg, -1))
            used += 1
        lp = torch.stack(lps).cuda().contiguous()
        res = eng.decode_retrieve_rerank(lp, [T] * B)
        for _ in range(3):
            eng.decode_retrieve_rerank(lp, [T] * B, want_text=False)
        regions = []
        for _ in range(3):      # a region is ~60 ms of synchronous calls: one host hiccup doubles it (seen in round 5) -> median of three
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.decode_retrieve_rerank(lp, [T] * B, want_text=False)
            torch.cuda.synchronize()
            regions.append((time.perf_counter() - t0) / args.steps)
        dt = sorted(regions)[1]
        row = {"case": name, "batch": B, "frames": T, "ms_per_batch": round(dt * 1e3, 3),
               "regions_ms": [round(r * 1e3, 3) for r in regions],
               "gate_failed": sum(r["use_ctc"] for r in res),
               "mean_candidates": round(sum(r["n_candidates"] for r in res) / B, 1),
               "mean_transcript_chars": round(sum(len(r["transcript"]) for r in res) / B, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
