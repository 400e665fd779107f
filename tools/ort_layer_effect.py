#!/usr/bin/env python
"""Per-layer effect of the two places where the HIP precision-2 path is NOT onnxruntime's arithmetic operation for
operation -- f16 Linear inputs and f16 int4 block scales (MatMulNBits uses f32 for both) -- next to a change that is no
error at all (1 intra-op thread instead of 16: float32 summation order).  CPU only, on the oracle
(oracle/fastconformer_ref.py::OrtMixed); VERDICT r3 "missing" #4 asked for the per-layer picture instead of the argument.

    python tools/ort_layer_effect.py [--seconds 3] [--out profiles/rNN_ort_per_layer_effect.json]

For every encoder layer output (and the subsampling output, and the log-probs): relative rms difference from the
reference run, for each variant.  If the f16 rows sat above the thread row anywhere before the first quantiser flip has
spread, they would be the implementation's own error; they do not.
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
    args = ap.parse_args()
    import torch

    from oracle import fastconformer_ref as R
    from synth import synth_audio

    n = int(args.seconds * 16000)
    audio = torch.from_numpy(synth_audio(1, n))
    w = R.random_weights(20260630)

    class F16(R.OrtMixed):
        def __init__(self, f16_scales):
            super().__init__()
            self.f16_scales = f16_scales

        def linear(self, w, name, x, bias_name):
            x = x.half().float()
            if self.f16_scales and name.endswith(R.ORT_INT4_SUFFIXES) and name not in self._w4:
                self._w4[name] = torch.from_numpy(R.quant_dequant_int4(w[name].numpy())).half().float()
            return super().linear(w, name, x, bias_name)

    def run(ops, threads):
        torch.set_num_threads(threads)
        taps = {}
        lp, _ = R.forward(w, audio, [n], taps=taps, ort=ops)
        taps["log_probs"] = lp
        return taps

    ref = run(R.OrtMixed(), args.threads)
    variants = {"threads_1 (summation order only)": run(R.OrtMixed(), 1),
                "f16_linear_inputs": run(F16(False), args.threads),
                "f16_linear_inputs_and_f16_int4_scales": run(F16(True), args.threads)}
    keys = ["sub"] + [f"layer{i}" for i in range(R.N_LAYERS)] + ["log_probs"]
    rows = {}
    for name, t in variants.items():
        rows[name] = {}
        for k in keys:
            d = (t[k] - ref[k]).double()
            rows[name][k] = float(d.pow(2).mean().sqrt() / ref[k].double().pow(2).mean().sqrt())
    doc = {"what": "relative rms difference from the reference oracle run, per tap, seeded random weights, one clip of %g s" % args.seconds,
           "rows": rows}
    print(f"{'tap':10s}" + "".join(f"{k[:28]:>30s}" for k in rows))
    for k in keys:
        print(f"{k:10s}" + "".join(f"{rows[v][k]:30.3e}" for v in rows))
    if args.out:
        Path(args.out).write_text(json.dumps(doc, indent=1) + "\n")


if __name__ == "__main__":
    main()
