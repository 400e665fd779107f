#!/usr/bin/env python
"""Streaming row: cost of the verse tracker's matching step (qv_tracker_match) per accumulated
text.  With --cpu-texts > 0 the CPU oracle's restatement of the reference loop is timed next to it
on the host (a reported baseline, like bench.py's cpu_baseline leg -- never part of the product path).

    python tools/tracker_bench.py [--steps 20] [--cpu-texts 8]
"""
from __future__ import annotations

import argparse
import json
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu-texts", type=int, default=8)
    args = ap.parse_args()

    import torch

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from oracle.tracker_ref import TrackerOracle

    eng = Engine(device=0, with_model=False, max_batch=16)
    tr = TrackerOracle()
    rng = random.Random(20260630)

    def texts_of(n, words):
        out = []
        while len(out) < n:
            v = rng.randrange(6236 - 8)
            w = " ".join(tr.o.verse_text(v + j) for j in range(8)).split()
            if len(w) >= words:
                out.append(" ".join(w[:words]))
        return out

    for words in (4, 12, 40):
        for batch in (1, 16, 64, 256):
            texts = texts_of(batch, words)
            lasts = [None] * batch
            eng.track_match(texts, lasts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.track_match(texts, lasts)
            dt = (time.perf_counter() - t0) / args.steps
            row = {"words": words, "mean_chars": round(sum(map(len, texts)) / batch, 1), "batch": batch,
                   "ms_per_call": round(dt * 1e3, 3), "us_per_text": round(dt * 1e6 / batch, 2)}
            if batch == 1:
                sample = texts_of(args.cpu_texts, words)
                t0 = time.perf_counter()
                for t in sample:
                    tr.best_raw(t, None)
                row["cpu_oracle_ms_per_text"] = round((time.perf_counter() - t0) * 1e3 / len(sample), 1)
            print(json.dumps(row), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
