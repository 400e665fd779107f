#!/usr/bin/env python
"""How far is each weight/activation treatment from the arithmetic the reference's model file runs?

    python tools/ort_delta.py [--seconds 3] [--out profiles/rNN_ort_semantics_delta.json]

Seeded random weights (the real file is absent), two ragged synthetic clips.  `ort` = oracle/fastconformer_ref.py
with OrtMixed(): MatMulNBits int4 (block 128, symmetric, FLOAT32 scales) on every Linear, DynamicQuantizeLinear +
ConvInteger (per-call uint8 activations, per-tensor int8 weights) on every Conv -- the reading of
experiments/c2c-direct-mixed/run.py:1-9 that quantises the most.  Reported: max |delta log-prob| over the valid
frames and the fraction of frames with the same argmax, for
  fp32                 the unquantised model (what the HIP fp16 path tracks to 4e-3)
  device_weights_fp32  the weights the HIP mixed engine holds (int4 with fp16 scales, per-channel int8 pointwise
                       convs), fp32 activations, no activation quantisation
  hip_fp16 / hip_mixed / hip_ort_mixed   the HIP forward itself under QV_PREC_FP16 / _MIXED_INT4_INT8 / _ORT_MIXED (only
                       where a GPU is visible); the last one runs the reference's arithmetic and lands on the oracle's own
                       noise floor (tools/ort_noise_floor.py)
Each ablation row switches on one piece of the onnxruntime arithmetic at a time.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch

    from oracle import fastconformer_ref as R
    from synth import synth_audio

    torch.set_num_threads(min(16, torch.get_num_threads()))
    n = int(args.seconds * 16000)
    lens = [n, n - 8000]
    audio = torch.from_numpy(synth_audio(2, n))
    audio[1, lens[1]:] = 0
    w = R.random_weights(20260630)
    lp_ort, T = R.forward(w, audio, lens, ort=R.OrtMixed())
    T = [int(t) for t in T]

    def cmp(lp):
        dd = torch.cat([(lp[b, : T[b]] - lp_ort[b, : T[b]]).flatten() for b in range(2)])
        same = sum(int((lp[b, : T[b]].argmax(-1) == lp_ort[b, : T[b]].argmax(-1)).sum()) for b in range(2)) / sum(T)
        return {"max_abs_delta_logprob": round(float(dd.abs().max()), 4), "rms_delta_logprob": round(float(dd.pow(2).mean().sqrt()), 5),
                "argmax_agreement": round(same, 4)}

    torch.set_num_threads(1)
    rows = {"oracle_itself_1_thread": cmp(R.forward(w, audio, lens, ort=R.OrtMixed())[0])}
    torch.set_num_threads(16)
    rows.update({"fp32": cmp(R.forward(w, audio, lens)[0]),
            "device_weights_fp32": cmp(R.forward(R.quantize_linear_weights(w), audio, lens)[0]),
            "ablation_int4_f32scale_only": cmp(R.forward(w, audio, lens, ort=R.OrtMixed(True, ()))[0]),
            "ablation_int4_plus_int8_pointwise_convs": cmp(R.forward(w, audio, lens, ort=R.OrtMixed(True, R.INT8_CONV_SUFFIXES))[0]),
            "ablation_all_convs_int8_no_int4": cmp(R.forward(w, audio, lens, ort=R.OrtMixed(False, "all"))[0])})
    if torch.cuda.is_available():
        import offline_tarteel_amd  # noqa: F401
        from offline_tarteel_amd.engine import Engine

        for name, prec in (("hip_fp16", 0), ("hip_mixed", 1), ("hip_ort_mixed", 2)):
            eng = Engine(device=0, with_model=True, seed=20260630, max_batch=2, max_samples=n, precision=prec)
            lp, Tg = eng.forward(audio.cuda().contiguous(), lens)
            assert list(Tg) == T
            rows[name] = cmp(lp.cpu())
            eng.close()
    doc = {"what": "max |delta log-prob| and argmax agreement against the onnxruntime-semantics oracle "
                   "(oracle/fastconformer_ref.py::OrtMixed: int4 MatMulNBits with f32 scales + dynamic uint8 activations / "
                   "per-tensor int8 weights on every Conv), seeded random weights, clips of %g s" % args.seconds,
           "frames": T, "rows": rows}
    print(json.dumps(doc, indent=1))
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
