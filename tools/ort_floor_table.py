#!/usr/bin/env python
"""QV_PREC_ORT_MIXED (the reference file's own int4 / int8 arithmetic) against its CPU restatement, per WEIGHT SET, next to the
restatement's own reproducibility floor on the same clips (tests/ort_floor.py) -- the table behind
tests/test_gpu_ort_mixed.py::test_weight_sets_stay_on_their_own_floor:

  random      seeded N(0, 1/fan_in) weights (the benchmark's): the worst case for rounding-boundary flips
  structured  rank-16 + 15 % i.i.d. matrices, CTC head x6 with a blank bias (peaked posteriors)
  damped      random weights with the residual branches' output matrices x0.25 and the CTC head x0.1: the set whose floor
              lies BELOW north_star's 1e-2, where the device is held to 1e-2 ABSOLUTE

    python tools/ort_floor_table.py --out gpurun_out/ort_floor_table.json
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--seed", type=int, default=20260630)
    a = ap.parse_args()
    import torch

    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from oracle import fastconformer_ref as R
    from ort_floor import delta, oracle_floor
    from synth import synth_audio

    spec = importlib.util.spec_from_file_location("convert_weights", str(ROOT / "tools" / "convert_weights.py"))
    C = importlib.util.module_from_spec(spec)
    sys.modules["convert_weights"] = C
    spec.loader.exec_module(C)
    shapes = C.weight_shapes(C._lib())
    lens = [48000, 30000]
    audio = torch.from_numpy(synth_audio(2, 48000, seed=5))
    audio[1, lens[1]:] = 0
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rows = {}
    for name, make in (("random", R.random_weights), ("structured", R.structured_weights), ("damped", R.damped_weights)):
        w = make(a.seed)
        lp_ref, t_ref = R.forward(w, audio, lens, ort=R.OrtMixed())
        T = t_ref.tolist()
        floor = oracle_floor(R, w, audio, lens, lp_ref, T)
        with tempfile.TemporaryDirectory() as td:
            path = Path(td) / f"{name}.qvw"
            C.write_qvw(path, {k: w[k].numpy() for k in shapes})
            eng = Engine(device=0, with_model=True, weights_path=str(path), precision=2, max_batch=2, max_samples=48000)
            try:
                lp, t = eng.forward(audio.cuda().contiguous(), lens)
                assert t == T
                mx, rms, same = delta(lp, lp_ref, T)
            finally:
                eng.close()
        peak = float(torch.cat([lp_ref[b, : T[b]].exp().max(-1).values for b in range(2)]).mean())
        rows[name] = {"device_vs_oracle": {"max": round(mx, 5), "rms": round(rms, 6), "argmax_agreement": round(same, 4)},
                      "oracle_vs_itself": {"max": round(floor["max"], 5), "rms": round(floor["rms"], 6), "argmax_agreement": round(floor["argmax"], 4),
                                           "rows": {k: [round(v[0], 5), round(v[1], 6)] for k, v in floor["rows"].items()}},
                      "mean_max_probability": round(peak, 4),
                      "floor_below_1e-2": floor["max"] < 1e-2,
                      "device_within_1e-2_absolute": mx <= 1e-2,
                      "device_within_1.5x_floor": mx <= 1.5 * floor["max"]}
        print(name, json.dumps(rows[name]), flush=True)
    doc = {"what": "QV_PREC_ORT_MIXED log-probs on the device vs the CPU restatement (oracle/fastconformer_ref.py OrtMixed), two ragged clips of 3 s, "
                   "per weight set, next to the restatement's distance from ITSELF under 1 thread / 1e-7 relative noise on its Linear inputs",
           "seed": a.seed, "weight_sets": rows}
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
