#!/usr/bin/env python
"""Clip-length sweep of the hot path on one GPU (5-30 s clips plus the mixed-length
batch), same engine and timing discipline as bench.py.  Writes one JSON document.

    python tools/sweep.py [--out profiles/rNN_sweep.json] [--steps 20] [--contexts 4]
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
    ap.add_argument("--out", default="")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--contexts", type=int, default=4)
    ap.add_argument("--precision", choices=("fp16", "mixed"), default="fp16")
    args = ap.parse_args()

    import offline_tarteel_amd  # noqa: F401  (first: sets GPU_MAX_HW_QUEUES before HIP initialises)
    import numpy as np
    import torch
    from offline_tarteel_amd.engine import Engine
    from synth import synth_audio

    B = args.batch
    nmax = 480000
    ncap = 534000   # 1.1x-slowed 30 s clips (TTA) still fit
    eng = Engine(device=0, with_model=True, seed=20260630, max_batch=B, max_samples=ncap, contexts=args.contexts,
                 precision=1 if args.precision == "mixed" else 0)
    base = synth_audio(B, nmax, seed=20260630)
    rng = np.random.default_rng(20260630)
    cases = [("5s", [80000] * B), ("10s", [160000] * B), ("20s", [320000] * B), ("30s", [480000] * B),
             ("mixed_5_30s", sorted(int(x) for x in rng.integers(80000, 480001, size=B)))]
    rows = []
    for name, lens in cases:
        n = max(lens)
        a = base[:, :n].copy()
        for b, ln in enumerate(lens):
            a[b, ln:] = 0.0
        audio = torch.from_numpy(a).cuda().contiguous()
        for _ in range(args.warmup):
            eng.predict_batch_async(audio, lens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.predict_batch_async(audio, lens)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        secs = sum(lens) / 16000.0
        rows.append({"case": name, "batch": B, "audio_seconds_per_batch": round(secs, 1),
                     "ms_per_batch": round(dt / args.steps * 1e3, 3),
                     "utterances_per_s": round(B * args.steps / dt, 1),
                     "audio_seconds_per_s": round(secs * args.steps / dt, 1),
                     "encoder_frames_max": eng.frames_for(n)})
        print(json.dumps(rows[-1]), flush=True)
    # BASELINE configs[4] flavour: TTA on 30 s clips with EVERY clip gated (worst case): the anchor batch,
    # then the 0.9x and 1.1x copies (GPU polyphase resampler) as two more batches, pipelined over the contexts
    lens = [480000] * B
    audio = torch.from_numpy(base[:, :480000].copy()).cuda().contiguous()

    def tta_step():
        eng.predict_batch_async(audio, lens)
        for f in (0.9, 1.1):
            cp = [eng.speed_perturb(audio[b], f) for b in range(B)]
            n = int(cp[0].numel())
            eng.predict_batch_async(torch.stack(cp).contiguous(), [n] * B)

    for _ in range(2):
        tta_step()
    torch.cuda.synchronize()
    nst = max(4, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(nst):
        tta_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows.append({"case": "tta_30s_all_gated", "batch": B, "audio_seconds_per_batch": 1920.0,
                 "ms_per_batch": round(dt / nst * 1e3, 3), "utterances_per_s": round(B * nst / dt, 1),
                 "audio_seconds_per_s": round(1920.0 * nst / dt, 1),
                 "note": "anchor + 0.9x + 1.1x passes of all 64 clips, two batched resampler launches per batch (qv_upfirdn_batch)"})
    print(json.dumps(rows[-1]), flush=True)
    doc = {"what": "c2c-direct-mixed hot path, one MI355X, synthetic clips resident in HBM, whole path per batch",
           "batches_in_flight": eng.contexts, "weights": args.precision, "steps": args.steps, "rows": rows}
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")
    eng.close()


if __name__ == "__main__":
    main()
