#!/usr/bin/env python
"""dev: the soak's setting with stage taps.  A four-context engine keeps three batches in flight; every batch also goes
through two one-context engines one after the other.  When their results differ, the stage taps of the two (both idle by
then) are compared from the front: the first one that differs names the kernel.

    QVERSE_DEBUG_TAPS=1 python tools/dev_ort_race.py [--batches 700] [--seed 13] [--precision 2]
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=700)
    ap.add_argument("--seed", type=int, default=13)
    ap.add_argument("--precision", type=int, default=2)
    ap.add_argument("--bg-precision", type=int, default=-1, help="precision of the four-context engine in the background (default: the same)")
    ap.add_argument("--bg-tree", default="", help="take the background engine (library + binding) from another checkout of the repo")
    ap.add_argument("--bg-torch", action="store_true", help="background load = torch matmuls on a side stream instead of an engine")
    args = ap.parse_args()
    os.environ["QVERSE_DEBUG_TAPS"] = "1"
    import numpy as np
    import torch

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from synth import synth_audio

    cap = 480000
    rng = np.random.default_rng(args.seed)
    pool = torch.from_numpy(synth_audio(64, cap, seed=args.seed)).cuda().contiguous()
    mk = lambda c: Engine(device=0, with_model=True, seed=5, precision=args.precision, max_batch=64, max_samples=cap, contexts=c)  # noqa: E731
    ea, eb = mk(1), mk(1)
    e4 = None
    BgEngine = Engine
    if args.bg_tree:
        import importlib
        import importlib.util
        pkg_dir = Path(args.bg_tree).resolve() / "offline-tarteel_amd"
        spec = importlib.util.spec_from_file_location("bg_pkg", str(pkg_dir / "__init__.py"), submodule_search_locations=[str(pkg_dir)])
        pkg = importlib.util.module_from_spec(spec)
        sys.modules["bg_pkg"] = pkg
        spec.loader.exec_module(pkg)
        mod = importlib.import_module("bg_pkg.engine")
        BgEngine = mod.Engine
    if not args.bg_torch:
        e4 = BgEngine(device=0, with_model=True, seed=5, precision=args.precision if args.bg_precision < 0 else args.bg_precision,
                    max_batch=64, max_samples=cap, contexts=4)
    side = torch.cuda.Stream()
    mats = [torch.randn(4096, 4096, device="cuda", dtype=torch.float16) for _ in range(2)]
    key = lambda r: (r["surah"], r["ayah"], r["ayah_end"], r["source"], r["score"], r["t_frames"], r["n_candidates"])  # noqa: E731

    def taps(e, B, lens, t):
        T = max(t)
        tm = [L // 160 + 1 for L in lens]
        l2 = [((x - 1) // 2 + 1 - 1) // 2 + 1 for x in tm]
        out = [("mel", e.forward_tap(0, 0, (B, max(tm), 80)), tm)]
        if args.precision == 2:
            out += [("c1 (k_sub01_ort)", e.forward_tap(6, 0, (B, max(l2), 20, 256)), l2),
                    ("c1p", e.forward_tap(7, 0, (B, max(l2), 20, 256)), l2),
                    ("c2", e.forward_tap(8, 0, (B, T, 10, 256)), t), ("c2p", e.forward_tap(9, 0, (B, T, 10, 256)), t)]
        out.append(("x0", e.forward_tap(1, 0, (B, T, 512)), t))
        for l in range(17):
            if args.precision == 2:
                out += [(f"ln_conv{l}", e.forward_tap(3, l, (B, T, 512)), t), (f"glu{l}", e.forward_tap(4, l, (B, T, 512)), t),
                        (f"dw{l}", e.forward_tap(5, l, (B, T, 512)), t)]
            out.append((f"x{l + 1}", e.forward_tap(2, l, (B, T, 512)), t))
        return out

    inflight, bad = [], 0
    try:
        for i in range(args.batches):
            B = int(rng.integers(1, 65))
            kind = rng.random()
            hi = 16000 if kind < 0.2 else 160000 if kind < 0.8 else cap
            lens = [int(x) for x in rng.integers(800, hi + 1, size=B)]
            a = pool[:B, : max(lens)].contiguous()
            for b, L in enumerate(lens):
                a[b, L:] = 0
            a0 = a.clone()
            if e4 is not None:
                inflight.append(e4.predict_batch_async(a, lens))
            else:
                with torch.cuda.stream(side):
                    for _ in range(6):
                        mats[0] @ mats[1]
            ra = ea.predict_batch(a, lens, want_text=False)
            rb = eb.predict_batch(a, lens, want_text=False)
            if [key(r) for r in ra] != [key(r) for r in rb]:
                bad += 1
                t = [r["t_frames"] for r in ra]
                torch.cuda.synchronize()
                ta, tb = taps(ea, B, lens, t), taps(eb, B, lens, t)
                first = []
                for (name, xa, nv), (_, xb, _) in zip(ta, tb):
                    d = [b for b in range(B) if not torch.equal(xa[b, : nv[b]], xb[b, : nv[b]])]
                    if d:
                        b0 = d[0]
                        n = int((xa[b0, : nv[b0]] != xb[b0, : nv[b0]]).sum())
                        first.append(f"{name}:{len(d)}utt,first {b0} len {lens[b0]} ({n}/{xa[b0, : nv[b0]].numel()})")
                first = first[:3] + (["..."] if len(first) > 3 else [])
                tm = [L // 160 + 1 for L in lens]
                fa, fb = ea.forward_tap(10, 0, (B, max(tm), 80)), eb.forward_tap(10, 0, (B, max(tm), 80))
                torch.cuda.synchronize()
                ea.predict_batch(a, lens, want_text=False)
                fa2 = ea.forward_tap(10, 0, (B, max(tm), 80))
                print("   audio tensor unchanged:", torch.equal(a, a0), "| first engine run again: log-mel equals its first run",
                      torch.equal(fa2, fa), "equals the second engine's", torch.equal(fa2, fb), flush=True)
                for b in range(B):
                    d = (fa[b, : tm[b]] != fb[b, : tm[b]]).nonzero()
                    if len(d):
                        print(f"   raw log-mel of utterance {b} (len {lens[b]}, {tm[b]} frames): {len(d)} values differ; first (frame, bin): "
                              f"{d[:8].tolist()} ... last {d[-1].tolist()};  first engine {fa[b][tuple(d[0])].item()!r} "
                              f"({fa[b][tuple(d[0])].view(torch.int32).item():#x}), second {fb[b][tuple(d[0])].item()!r}", flush=True)
                        break
                print("MISMATCH at batch", i, "B", B, "max_len", max(lens), "differing stages:", first, flush=True)
            if len(inflight) == 4:
                e4.wait(inflight.pop(0))
    finally:
        torch.cuda.synchronize()
        if e4 is not None:
            e4.close()
        ea.close(); eb.close()
    print(f"{args.batches} batches, {bad} mismatching between the two one-context engines")


if __name__ == "__main__":
    main()
