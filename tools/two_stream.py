#!/usr/bin/env python
"""dev experiment: do two batches in flight on two HIP streams (two engine contexts) overlap the
latency-bound post-logits kernels of one with the forward of the other?"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch

import offline_tarteel_amd  # noqa: F401
from offline_tarteel_amd.engine import Engine
from synth import synth_audio

B, n = 64, 160000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
audio = [torch.from_numpy(synth_audio(B, n, seed=20260630 + i)).cuda().contiguous() for i in range(nctx)]
engs = [Engine(device=0, with_model=True, seed=20260630, max_batch=B, max_samples=n) for _ in range(nctx)]
streams = [torch.cuda.Stream() for _ in range(nctx)]
lengths = [n] * B


def run(k, ctxs):
    for i in range(k):
        c = i % ctxs
        with torch.cuda.stream(streams[c]):
            engs[c].predict_batch_async(audio[c], lengths)


for ctxs in (1, nctx, 1, nctx):
    run(4, ctxs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, ctxs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"contexts={ctxs}: {dt / steps * 1e3:.3f} ms/step  {B * steps / dt:.0f} utt/s")
