#!/usr/bin/env python
"""dev: dump the log-probs of a ragged seeded batch (compare two builds / env switches bitwise)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import offline_tarteel_amd  # noqa
from offline_tarteel_amd.engine import Engine
from synth import synth_audio
lens = [160000, 80000, 123457, 47777, 159999, 16000]
a = torch.from_numpy(synth_audio(len(lens), 160000, seed=5))
for b, n in enumerate(lens):
    a[b, n:] = 0
eng = Engine(device=0, with_model=True, seed=3, max_batch=8, max_samples=160000)
lp, t = eng.forward(a.cuda().contiguous(), lens)
torch.save({"lp": lp.cpu(), "t": t}, sys.argv[1])
print("saved", sys.argv[1], t)
