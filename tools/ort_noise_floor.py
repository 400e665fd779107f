#!/usr/bin/env python
"""How reproducible is the reference's quantised arithmetic -- against ITSELF?

    python tools/ort_noise_floor.py [--seconds 3] [--out profiles/rNN_ort_noise_floor.json]

north_star asks for CTC log-probs within 1e-2 of the reference's c2c-direct-mixed model.  That model quantises the
activations of every Conv per call (DynamicQuantizeLinear: uint8, range taken from the tensor's own min / max), i.e.
it puts ~60 rounding discontinuities between the audio and the log-probs.  A perturbation far below any tolerance
(float32 summation order, 1 ulp on an input) moves a handful of values across a rounding boundary, and from there the
difference is carried and amplified like any other quantisation noise.  This tool measures that floor on the CPU
oracle (oracle/fastconformer_ref.py::OrtMixed, seeded random weights -- the real file is absent), by comparing the
oracle with itself under changes that are NOT errors:

  threads_1_vs_N     the same code with 1 and with N intra-op threads (float32 GEMM blocking / summation order)
  ulp_noise_1e-7     every Linear input multiplied by (1 + 1e-7 u), u ~ U(-1, 1): < 1 ulp of float32
  f16_linear_inputs  every Linear input rounded to float16 (what an f16-operand MFMA GEMM sees; the quantisers and the
                     integer convolutions stay exact) -- the design point of the HIP path's QV_PREC_ORT_MIXED
  f16_int4_scales    additionally the int4 block scales and the dequantised weights rounded to float16

Reported per row: max |delta log-prob|, rms delta, fraction of frames with the same argmax.  Any implementation of
this arithmetic -- onnxruntime itself on another machine included -- differs from another by about the first two
rows; a tolerance below that floor cannot be met by anything but bit-identical summation order.
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
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--out", default="")
    ap.add_argument("--structured", action="store_true",
                    help="low-rank + noise weight matrices and a peaked, blank-biased CTC head (fastconformer_ref.structured_weights) "
                         "instead of i.i.d. weights: does the floor drop when activations and posteriors have structure?")
    ap.add_argument("--head-gain", type=float, default=6.0, help="with --structured: factor on the CTC head's weight")
    args = ap.parse_args()
    import numpy as np
    import torch

    from oracle import fastconformer_ref as R
    from synth import synth_audio

    n = int(args.seconds * 16000)
    lens = [n, n - 8000]
    audio = torch.from_numpy(synth_audio(2, n))
    audio[1, lens[1]:] = 0
    w = R.structured_weights(20260630, head_gain=args.head_gain) if args.structured else R.random_weights(20260630)
    torch.set_num_threads(args.threads)
    lp_ref, T = R.forward(w, audio, lens, ort=R.OrtMixed())
    T = [int(t) for t in T]

    def cmp(lp):
        d = torch.cat([(lp[b, : T[b]] - lp_ref[b, : T[b]]).flatten() for b in range(2)])
        same = sum(int((lp[b, : T[b]].argmax(-1) == lp_ref[b, : T[b]].argmax(-1)).sum()) for b in range(2)) / sum(T)
        return {"max_abs_delta_logprob": round(float(d.abs().max()), 4), "rms_delta_logprob": round(float(d.pow(2).mean().sqrt()), 5),
                "argmax_agreement": round(same, 4)}

    class Noise(R.OrtMixed):
        def __init__(self, eps):
            super().__init__()
            self.eps, self.g = eps, torch.Generator().manual_seed(1)

        def linear(self, w, name, x, bias_name):
            x = x * (1 + self.eps * (torch.rand(x.shape, generator=self.g) * 2 - 1))
            return super().linear(w, name, x, bias_name)

    class F16(R.OrtMixed):
        def __init__(self, f16_scales):
            super().__init__()
            self.f16_scales = f16_scales

        def linear(self, w, name, x, bias_name):
            x = x.half().float()
            if self.f16_scales and name.endswith(R.ORT_INT4_SUFFIXES):
                if name not in self._w4:   # (q - 8) * half(scale), product rounded to half: csrc/qv_gemm_dequant.h::dequant8
                    self._w4[name] = torch.from_numpy(R.quant_dequant_int4(w[name].numpy())).half().float()
            return super().linear(w, name, x, bias_name)

    rows = {}
    rows["rerun_same_threads"] = cmp(R.forward(w, audio, lens, ort=R.OrtMixed())[0])
    torch.set_num_threads(1)
    rows[f"threads_1_vs_{args.threads}"] = cmp(R.forward(w, audio, lens, ort=R.OrtMixed())[0])
    torch.set_num_threads(args.threads)
    rows["ulp_noise_1e-7"] = cmp(R.forward(w, audio, lens, ort=Noise(1e-7))[0])
    rows["f16_linear_inputs"] = cmp(R.forward(w, audio, lens, ort=F16(False))[0])
    rows["f16_linear_inputs_f16_int4_scales"] = cmp(R.forward(w, audio, lens, ort=F16(True))[0])
    doc = {"what": "self-consistency of the onnxruntime-semantics oracle (OrtMixed: int4 MatMulNBits + DynamicQuantizeLinear / "
                   "ConvInteger on every Conv) under changes that are not errors; seeded random weights, two clips of %g s, "
                   "reference = the oracle itself with %d threads" % (args.seconds, args.threads),
           "weights": f"structured (rank-16 + 15 % i.i.d., CTC head x{args.head_gain:g}, blank bias +3)" if args.structured else "i.i.d. seeded",
           "posterior_peak": {"mean_max_prob": round(float(torch.cat([lp_ref[b, : T[b]].exp().max(-1).values for b in range(2)]).mean()), 4),
                              "blank_argmax_fraction": round(float(torch.cat([(lp_ref[b, : T[b]].argmax(-1) == 1024).float() for b in range(2)]).mean()), 4)},
           "frames": T, "rows": rows}
    print(json.dumps(doc, indent=1))
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
