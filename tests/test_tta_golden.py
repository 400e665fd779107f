"""TTA wrapper (a15) against cases generated from the unmodified reference predict()
(tests/golden/gen_tta_golden.py -> tta_cases.json): the host decision rule of the plugin, and the
oracle's per-pass decisions on the same synthetic log-probs (CPU)."""

import json

import pytest
import torch

from synth import synth_logits


@pytest.fixture(scope="module")
def tta_cases(golden_dir):
    return json.loads((golden_dir / "tta_cases.json").read_text(encoding="utf-8"))


def decide(plugin, p09, anchor, p11):
    """gate + rule exactly as plugin.tta_device_batch applies them per clip"""
    if anchor["score"] >= plugin.CONFIDENCE_SKIP_THRESHOLD:
        return anchor
    return plugin._tta_combine(p09, anchor, p11)


def same_dict(got, want, score_rel=0.0):
    got = {k: ([list(x) for x in v] if k == "tta_preds" else v) for k, v in got.items() if k != "candidates"}
    want = dict(want)
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k, w in want.items():
        g = got[k]
        if k == "score":
            assert abs(g - w) <= score_rel * max(abs(w), 1e-3), (k, g, w)
        elif k == "tta_scores":
            assert len(g) == len(w) and all(abs(a - b) <= score_rel * max(abs(b), 1e-3) for a, b in zip(g, w)), (g, w)
        else:
            assert g == w, (k, g, w)


def test_decision_rule_equals_reference_on_scripted_cases(tta_cases):
    from offline_tarteel_amd import plugin

    assert plugin.CONFIDENCE_SKIP_THRESHOLD == tta_cases["gate"]
    assert len(tta_cases["scripted"]) >= 10
    for c in tta_cases["scripted"]:
        got = decide(plugin, dict(c["p09"]), dict(c["anchor"]), dict(c["p11"]))
        same_dict(got, c["out"])


def lp_of(r):
    lg = synth_logits(r["ids"], r["T"], seed=r["seed"], noise=r["noise"], boost=r["boost"], rep=r["rep"])
    return torch.log_softmax(torch.from_numpy(lg), -1)


def oracle_pass(oracle, recipe):
    r = oracle.predict_logprobs(lp_of(recipe).numpy())
    if not r["surah"]:
        return {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0, "transcript": r["transcript"]}
    return {"surah": r["surah"], "ayah": r["ayah"], "ayah_end": r["ayah_end"] or r["ayah"], "score": r["score_raw"],
            "transcript": r["transcript"], "source": r["source"]}


def test_oracle_passes_and_rule_on_logprob_cases(oracle, tta_cases):
    """every pass (0.9x, anchor, 1.1x log-probs) through the oracle, UNROUNDED scores as
    c2c-direct-mixed-tta/run.py:82-109 returns them, then the rule"""
    from offline_tarteel_amd import plugin

    for c in tta_cases["logprob"]:
        passes = [oracle_pass(oracle, r) for r in c["recipes"]]
        for got, want in zip(passes, c["per_pass"]):
            same_dict(got, want, score_rel=1e-9)
        same_dict(decide(plugin, *passes), c["out"], score_rel=1e-9)
