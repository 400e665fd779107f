"""Golden cases for the TTA wrapper (a15), generated from the UNMODIFIED reference
experiments/c2c-direct-mixed-tta/run.py::predict (lines 117-149).

Run ONLY in the build container (needs /root/reference):

    PYTHONHASHSEED=0 python tests/golden/gen_tta_golden.py

The reference's `predict` is driven with two of its own hooks replaced by scripted data sources
(the repo's mocking idiom, tests/test_streaming_pipeline.py:34-77): `load_audio` returns a seeded
array, and
  * "scripted" cases replace `_predict_one` by a table keyed on the clip length (the anchor, the
    resample_poly(x, 9, 10) copy and the resample_poly(x, 11, 10) copy have three different lengths):
    they pin the decision rule itself -- the 0.5 gate on the unrounded score, the majority over
    (surah, ayah) in the order [0.9x, anchor, 1.1x], the best-score pick and its tie order, the keys
    `tta`, `tta_preds`, `tta_scores`;
  * "logprob" cases replace only `_log_probs_from_audio` (the onnxruntime call) by the tests'
    synthetic log-prob recipe per clip length, so the reference's own `_log_probs_to_pred` (greedy
    decode, retrieval, gate, CTC rerank, UNROUNDED score) runs on each of the three passes.

Writes tests/golden/tta_cases.json (inputs + expected outputs; no reference source text).
"""

from __future__ import annotations

import importlib.util
import json
import os
import random
import sys
import types
from pathlib import Path

if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ref_import import REF, _Model, indel_ratio  # noqa: E402
from synth import synth_audio, synth_logits  # noqa: E402


def load_tta():
    lev = types.ModuleType("Levenshtein")
    lev.ratio = indel_ratio
    sys.modules["Levenshtein"] = lev
    for name in ("librosa", "soundfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))
    spec = importlib.util.spec_from_file_location("_tta_ref", str(REF / "experiments/c2c-direct-mixed-tta/run.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cd = mod._cdm._cd
    cd._model = _Model()
    cd._db = cd.QuranDB()
    mod._cdm._ensure_ort = lambda: None
    return mod


def clean(d):
    out = {}
    for k, v in d.items():
        if k == "candidates":
            continue
        if k == "tta_preds":
            v = [list(x) for x in v]
        out[k] = v
    return out


def main():
    tta = load_tta()
    cd = tta._cdm._cd
    tok = cd._model.tokenizer
    n = 24000
    audio = synth_audio(1, n, seed=77)[0]
    n09 = len(tta._speed_perturb(audio, 0.9))
    n11 = len(tta._speed_perturb(audio, 1.1))
    assert len({n, n09, n11}) == 3
    cd.load_audio = lambda path: audio

    def P(s, a, e, score, src="ctc", tr="x"):
        if s == 0:
            return {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0, "transcript": tr, "candidates": []}
        return {"surah": s, "ayah": a, "ayah_end": e, "score": score, "transcript": tr, "source": src}

    scripted_in = [
        ("anchor_confident", P(1, 1, 1, 0.3), P(1, 2, 2, 0.5, "text"), P(1, 3, 3, 0.9)),
        ("anchor_just_below_gate", P(2, 2, 2, 0.1), P(1, 2, 2, 0.4999999999), P(1, 2, 3, 0.2)),
        ("majority_09_anchor", P(36, 1, 5, 0.01), P(36, 1, 3, 0.2), P(36, 2, 2, 0.4)),
        ("majority_09_11_over_anchor", P(114, 1, 6, 0.19), P(56, 36, 36, 0.0), P(114, 1, 1, 0.02)),
        ("all_agree", P(1, 7, 7, 0.004), P(1, 7, 7, 0.003), P(1, 7, 7, 0.2)),
        ("score_pick_11", P(2, 1, 1, 0.1), P(3, 1, 1, 0.2), P(4, 1, 1, 0.3)),
        ("score_pick_tie_first_wins", P(2, 1, 1, 0.25), P(3, 1, 1, 0.25), P(4, 1, 1, 0.1)),
        ("score_pick_anchor", P(2, 1, 1, 0.1), P(3, 1, 1, 0.45), P(4, 1, 1, 0.3)),
        ("empty_anchor_two_empties_majority", P(0, 0, None, 0.0), P(0, 0, None, 0.0), P(5, 5, 5, 0.7)),
        ("empty_anchor_score_pick", P(9, 9, 9, 0.05), P(0, 0, None, 0.0), P(5, 5, 5, 0.7)),
        ("majority_same_ayah_different_end", P(55, 1, 4, 0.013), P(55, 1, 1, 0.4), P(67, 1, 4, 0.3)),
    ]
    scripted = []
    for name, p09, anchor, p11 in scripted_in:
        table = {n09: dict(p09), n: dict(anchor), n11: dict(p11)}
        tta._predict_one = lambda a, table=table: dict(table[len(a)])
        got = tta.predict("unused.wav")
        scripted.append({"name": name, "p09": clean(p09), "anchor": clean(anchor), "p11": clean(p11), "out": clean(got)})
        print(name, "->", {k: got.get(k) for k in ("surah", "ayah", "score", "tta")})

    # ---- logprob cases: the reference's own per-pass decision on synthetic log-probs --------------------
    spec = importlib.util.spec_from_file_location("_tta_ref2", str(REF / "experiments/c2c-direct-mixed-tta/run.py"))
    del tta._predict_one
    tta2 = load_tta()
    tta2._cdm._cd.load_audio = lambda path: audio
    db = tta2._cdm._cd._db

    def ids_of(text):
        return [int(i) for i in tok.text_to_ids(text)]

    def verse(s, a):
        return db.get_verse(s, a)

    def corrupt(ids, rate, seed):
        r = random.Random(seed)
        out = []
        for i in ids:
            x = r.random()
            if x < rate / 2:
                continue
            out.append(r.randrange(1, 1024) if x < rate else i)
        return out

    def recipe(ids, T, seed, noise, boost, rep):
        return {"ids": ids, "T": T, "seed": seed, "noise": noise, "boost": boost, "rep": rep}

    v1 = ids_of(verse(103, 2)["text_clean"])
    v2 = ids_of(verse(112, 2)["text_clean"])
    v3 = ids_of(" ".join(verse(114, a)["text_clean"] for a in range(1, 4)))
    lp_in = [
        # anchor recognised by the text gate with a high score: returned as is, no perturbed pass consulted
        ("lp_anchor_text_confident", recipe(v2, 24, 11, 1.0, 8.0, 2), recipe(v2, 22, 12, 1.0, 8.0, 2), recipe(v2, 27, 13, 1.0, 8.0, 2)),
        # anchor is noise below the gate, the perturbed passes disagree -> best unrounded score (a perturbed pass wins)
        ("lp_score_pick_perturbed_beats_noise_anchor", recipe(corrupt(v1, 0.3, 1), 44, 21, 1.0, 7.0, 2), recipe([], 48, 22, 1.0, 0.0, 1),
         recipe(corrupt(v1, 0.3, 2), 53, 23, 1.0, 7.0, 2)),
        # three different answers -> best unrounded score
        ("lp_score_pick", recipe(corrupt(v3, 0.35, 3), 80, 31, 1.0, 6.0, 2), recipe(corrupt(v1, 0.45, 4), 48, 32, 1.6, 5.0, 2),
         recipe(corrupt(v2, 0.5, 5), 30, 33, 1.6, 5.0, 2)),
    ]
    lp_cases = []
    for name, r09, r10, r11 in lp_in:
        table = {n09: r09, n: r10, n11: r11}

        def lp_from_audio(a, table=table):
            r = table[len(a)]
            lg = torch.from_numpy(synth_logits(r["ids"], r["T"], seed=r["seed"], noise=r["noise"], boost=r["boost"], rep=r["rep"]))
            return torch.log_softmax(lg, -1).numpy()

        tta2._log_probs_from_audio = lp_from_audio
        per_pass = [clean(tta2._log_probs_to_pred(lp_from_audio(np.zeros(k, np.float32)))) for k in (n09, n, n11)]
        got = tta2.predict("unused.wav")
        lp_cases.append({"name": name, "recipes": [r09, r10, r11], "per_pass": per_pass, "out": clean(got)})
        print(name, [(p["surah"], p["ayah"], round(p["score"], 4)) for p in per_pass], "->",
              {k: got.get(k) for k in ("surah", "ayah", "ayah_end", "score", "tta")})

    doc = {"gate": tta.CONFIDENCE_SKIP_THRESHOLD, "clip_samples": n, "n09": n09, "n11": n11,
           "scripted": scripted, "logprob": lp_cases}
    (HERE / "tta_cases.json").write_text(json.dumps(doc, ensure_ascii=False, separators=(",", ":")), encoding="utf-8")
    print("wrote tta_cases.json", (HERE / "tta_cases.json").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
