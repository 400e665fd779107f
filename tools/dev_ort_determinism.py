#!/usr/bin/env python
"""dev: is the precision-2 forward reproducible run to run?  The same ragged batch through the same engine N times, log-probs
and the stage taps compared bit for bit with the first run, while a second engine keeps the GPU busy on other streams (the
soak's setting).  Reports, per tap, in how many repetitions it differed and the first stage that did.

    python tools/dev_ort_determinism.py [--reps 60] [--precision 2] [--batch 53] [--max-len 15546] [--quiet-gpu]
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--precision", type=int, default=2)
    ap.add_argument("--batch", type=int, default=53)
    ap.add_argument("--max-len", type=int, default=15546)
    ap.add_argument("--quiet-gpu", action="store_true", help="no second engine in the background")
    ap.add_argument("--contexts", type=int, default=1)
    ap.add_argument("--history", action="store_true",
                    help="before every repetition run ANOTHER batch (shape and content change from repetition to repetition): "
                         "does the probe batch depend on what the engine processed before it?")
    args = ap.parse_args()
    os.environ["QVERSE_DEBUG_TAPS"] = "1"
    import numpy as np
    import torch

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from synth import synth_audio

    rng = np.random.default_rng(5)
    B = args.batch
    lens = [int(x) for x in rng.integers(800, args.max_len + 1, size=B)]
    n = max(lens)
    a = torch.from_numpy(synth_audio(B, n, seed=3)).cuda().contiguous()
    for b, L in enumerate(lens):
        a[b, L:] = 0
    eng = Engine(device=0, with_model=True, seed=5, precision=args.precision, max_batch=64, max_samples=480000, contexts=args.contexts)
    bg = None
    if not args.quiet_gpu:
        bg = Engine(device=0, with_model=True, seed=5, precision=args.precision, max_batch=64, max_samples=480000, contexts=4)
        bga = torch.from_numpy(synth_audio(64, 160000, seed=9)).cuda().contiguous()
        bgl = [160000] * 64

    def taps():
        lp, t = eng.forward(a, lens)
        torch.cuda.synchronize()
        out = {"logprobs": torch.cat([lp[b, : t[b]].flatten() for b in range(B)]).clone()}
        T = max(t)
        if args.precision == 2:
            tm = [L // 160 + 1 for L in lens]
            out["mel"] = eng.forward_tap(0, 0, (B, max(tm), 80)).clone()
            l1 = [(x - 1) // 2 + 1 for x in tm]; l2 = [(x - 1) // 2 + 1 for x in l1]
            out["c1"] = eng.forward_tap(6, 0, (B, max(l2), 20, 256)).clone()
            out["c1p"] = eng.forward_tap(7, 0, (B, max(l2), 20, 256)).clone()
            out["c2"] = eng.forward_tap(8, 0, (B, T, 10, 256)).clone()
            out["c2p"] = eng.forward_tap(9, 0, (B, T, 10, 256)).clone()
        out["x0"] = eng.forward_tap(1, 0, (B, T, 512)).clone()
        for l in (0, 1, 8, 16):
            if args.precision == 2:
                out[f"ln_conv{l}"] = eng.forward_tap(3, l, (B, T, 512)).clone()
                out[f"glu{l}"] = eng.forward_tap(4, l, (B, T, 512)).clone()
                out[f"dw{l}"] = eng.forward_tap(5, l, (B, T, 512)).clone()
            out[f"x{l + 1}"] = eng.forward_tap(2, l, (B, T, 512)).clone()
        return out, t

    hist = []
    if args.history:
        pool = torch.from_numpy(synth_audio(64, 480000, seed=21)).cuda().contiguous()
        for r in range(args.reps + 1):
            hb = int(rng.integers(1, 65))
            hi = [16000, 160000, 480000][r % 3]
            hl = [int(x) for x in rng.integers(800, hi + 1, size=hb)]
            hist.append((hb, hl))
    try:
        if hist:
            hb, hl = hist[-1]
            eng.predict_batch(pool[:hb, : max(hl)].contiguous(), hl, want_text=False)
        ref, t = taps()
        names = list(ref)
        bad = {k: 0 for k in names}
        first = {}
        for r in range(args.reps):
            if bg is not None:
                for _ in range(2):
                    bg.predict_batch_async(bga, bgl)
            if hist:
                hb, hl = hist[r]
                eng.predict_batch(pool[:hb, : max(hl)].contiguous(), hl, want_text=False)
            got, _ = taps()
            for k in names:
                same = torch.equal(got[k], ref[k])
                # padding rows of the dense conv taps are not defined: compare valid frames only where the tap is dense
                if not same and k in ("c1", "c1p", "c2", "c2p", "mel"):
                    same = True
                    for b in range(B):
                        nv = {"mel": lens[b] // 160 + 1, "c1": None, "c1p": None, "c2": t[b], "c2p": t[b]}[k]
                        if nv is None:
                            tm = lens[b] // 160 + 1
                            nv = ((tm - 1) // 2 + 1 - 1) // 2 + 1
                        same = same and torch.equal(got[k][b, :nv], ref[k][b, :nv])
                if not same:
                    bad[k] += 1
                    first.setdefault(r, k)
        print(f"precision {args.precision}, contexts {args.contexts}, B {B}, frames {min(t)}..{max(t)}, {args.reps} repetitions, "
              f"background load {'off' if bg is None else 'on'}, other batches in between {'yes' if hist else 'no'}")
        for k in names:
            print(f"  {k:10s} differed in {bad[k]} repetitions")
        print("  first differing stage per bad repetition:", sorted(set(first.values()), key=names.index))
    finally:
        eng.close()
        if bg is not None:
            bg.close()


if __name__ == "__main__":
    main()
