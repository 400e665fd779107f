#!/usr/bin/env python
"""What the post-logits kernels are handed on the TTA / 30 s workload (BASELINE configs[4]): per pass (0.9x, anchor,
1.1x) the transcript lengths, candidate counts, CTC leaders (one alpha recursion each) with their state counts, and
the host time of tta_start's Python half.  The candidate lists are rebuilt on the host with the oracle (test
infrastructure: this is a measurement tool, not the product path).

    python tools/diag_tta_work.py [--clips 8] [--seconds 30]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    a = ap.parse_args()
    import numpy as np
    import torch

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from oracle.oracle import Oracle
    from synth import synth_audio

    n = int(a.seconds * 16000)
    B = a.clips
    cap = int(n * 1.1) + 1600
    eng = Engine(device=0, with_model=True, seed=20260630, max_batch=64, max_samples=cap, contexts=1)
    orc = Oracle()
    audio = torch.from_numpy(synth_audio(64, n, seed=20260630)).cuda().contiguous()
    for name, f in (("anchor", 1.0), ("0.9x", 0.9), ("1.1x", 1.1)):
        clips = [eng.speed_perturb(audio[i].contiguous(), f) for i in range(B)]
        lens = [int(c.numel()) for c in clips]
        rows = torch.zeros((B, max(lens)), dtype=torch.float32, device="cuda")
        for i, c in enumerate(clips):
            rows[i, : lens[i]] = c
        res = eng.predict_batch(rows, lens, want_text=True)
        T = res[0]["t_frames"]
        st = {"pass": name, "t_frames": T, "chars": [], "n_cand": [], "n_lead": [], "use_ctc": 0,
              "lead_states_hist": {}, "sum_T_S": 0, "sum_T_S_padded": 0}
        for r in res:
            tx = r["transcript"]
            st["chars"].append(len(tx))
            st["use_ctc"] += int(r["use_ctc"])
            st["n_cand"].append(r["n_candidates"])
            cs, cp, sc, m = orc.build_candidates(tx)
            lead = {}
            for s, p in zip(cs.tolist(), cp.tolist()):
                L = len(orc.token_ids(s, p))
                if L > 0 and 2 * L + 1 <= T:
                    lead[s] = max(lead.get(s, 0), 2 * L + 1)
            st["n_lead"].append(len(lead))
            for S in lead.values():
                b = (S + 63) // 64 * 64
                st["lead_states_hist"][b] = st["lead_states_hist"].get(b, 0) + 1
                st["sum_T_S"] += T * S
                st["sum_T_S_padded"] += T * b
        st["lead_states_hist"] = dict(sorted(st["lead_states_hist"].items()))
        print(json.dumps(st), flush=True)
        print("  sample transcripts:", [r["transcript"][:40] for r in res[:3]], flush=True)
    # host half of the TTA wrapper: how long does tta_start's Python take for a batch of 64 gated clips?
    from offline_tarteel_amd.plugin import tta_finish, tta_start

    eng.close()
    eng = Engine(device=0, with_model=True, seed=20260630, max_batch=64, max_samples=cap, contexts=3)
    lengths = [n] * 64
    for it in range(3):
        ctx = eng.predict_batch_async(audio, lengths)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = tta_start(eng, audio, lengths, want_text=False, anchor_ctx=ctx)
        t1 = time.perf_counter()
        out = tta_finish(eng, st)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(json.dumps({"iter": it, "tta_start_host_ms": round((t1 - t0) * 1e3, 2), "tta_finish_ms": round((t2 - t1) * 1e3, 2),
                          "gated": sum(1 for r in out if "tta" in r)}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
