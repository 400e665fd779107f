"""GPU diagnostic (not collected by pytest): HIP forward vs the fp32 PyTorch restatement, stage by
stage.  Lives under tests/ because it uses the oracle.    python tests/diag_forward.py"""
import os, sys, time
os.environ["QVERSE_DEBUG_TAPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import offline_tarteel_amd
from offline_tarteel_amd.engine import Engine
from oracle import fastconformer_ref as R
from synth import synth_audio

seed = 7
lens = [48000, 30000, 17777]
audio = torch.from_numpy(synth_audio(3, 48000))
for b, n in enumerate(lens):
    audio[b, n:] = 0
t0 = time.time(); w = R.random_weights(seed); print("ref weights", time.time() - t0)
taps = {}
t0 = time.time(); lp_ref, T_ref = R.forward(w, audio, lens, taps=taps); print("ref fwd", time.time() - t0)
t0 = time.time(); eng = Engine(device=0, with_model=True, seed=seed, max_batch=4, max_samples=80000); print("engine", time.time() - t0)
lp, T = eng.forward(audio.cuda().contiguous(), lens)
torch.cuda.synchronize()
print("T", T, T_ref.tolist())
def cmp(name, a, b, lens_t):
    a = a.float().cpu(); b = b.float().cpu()
    for i, n in enumerate(lens_t):
        d = (a[i, :n] - b[i, :n]).abs()
        print(f"  {name}[{i}] max|d|={d.max():.4e} mean|d|={d.mean():.4e} ref_absmean={b[i,:n].abs().mean():.3e}")
tm = [n // 160 + 1 for n in lens]
mel = eng.forward_tap(0, 0, (3, max(tm), 80)); cmp("mel", mel, taps["mel"], tm)
Tm = max(T)
sub = eng.forward_tap(1, 0, (3, Tm, 512)); cmp("sub", sub, taps["sub"] * (512 ** 0.5), T)
for l in (0, 1, 4, 8, 16):
    x = eng.forward_tap(2, l, (3, Tm, 512)); cmp(f"layer{l}", x, taps[f"layer{l}"], T)
cmp("logprobs", lp, lp_ref, T)
am = [(lp[i, :T[i]].argmax(-1).cpu() == lp_ref[i, :T[i]].argmax(-1)).float().mean().item() for i in range(3)]
print("argmax agreement", am)
# batch invariance
lp1, T1 = eng.forward(audio[1:2, :30000].cuda().contiguous(), [30000])
print("batch invariance max|d|", (lp1[0, :T1[0]] - lp[1, :T1[0]]).abs().max().item())
# timing at B=64 10 s
eng.close()
eng = Engine(device=0, with_model=True, seed=seed, max_batch=64, max_samples=160000)
a = torch.from_numpy(synth_audio(64, 160000)).cuda()
ln = [160000] * 64
for _ in range(2): eng.forward(a, ln)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): eng.forward(a, ln)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"forward B=64 10s: {dt*1e3:.2f} ms -> {64/dt:.0f} utt/s, {64*28.5e9/dt/1e12:.1f} TFLOP/s algorithmic")
res = eng.predict_batch(a, ln)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): res = eng.predict_batch(a, ln)
torch.cuda.synchronize(); dt = (time.time() - t0) / 3
print(f"predict_batch B=64 10s: {dt*1e3:.2f} ms -> {64/dt:.0f} utt/s")
print(res[0])
