"""TTA (a15, BASELINE configs[4]) on the GPU: the reference-generated log-prob cases through the C ABI, and the
per-rank slice of configs[4] at full size -- 64 clips x 30 s through plugin.tta_device_batch -- against the
oracle run on the HIP path's own log-probs."""

import json

import numpy as np
import pytest
import torch

from synth import synth_audio, synth_logits

pytestmark = pytest.mark.gpu


def lp_of(r):
    lg = synth_logits(r["ids"], r["T"], seed=r["seed"], noise=r["noise"], boost=r["boost"], rep=r["rep"])
    return torch.log_softmax(torch.from_numpy(lg), -1)


def test_logprob_cases_through_the_c_abi(golden_dir):
    from offline_tarteel_amd import plugin
    from offline_tarteel_amd.engine import Engine
    from test_tta_golden import decide, same_dict

    cases = json.loads((golden_dir / "tta_cases.json").read_text(encoding="utf-8"))["logprob"]
    eng = Engine(device=0, with_model=False, max_batch=4)
    try:
        for c in cases:
            lps = [lp_of(r) for r in c["recipes"]]
            t_max = max(x.shape[0] for x in lps)
            batch = torch.full((3, t_max, 1025), -50.0)
            for b, x in enumerate(lps):
                batch[b, : x.shape[0]] = x
            res = eng.decode_retrieve_rerank(batch.cuda().contiguous(), [x.shape[0] for x in lps])
            passes = [plugin._to_dict(r, False) for r in res]
            for got, want in zip(passes, c["per_pass"]):
                # text scores are bit-exact, CTC scores exp(-loss/L) carry the fp32 recursion's 1e-3
                same_dict(got, want, score_rel=0.0 if want.get("source") == "text" else 1e-3)
            same_dict(decide(plugin, *passes), c["out"], score_rel=1e-3)
    finally:
        eng.close()


def test_configs4_rank_slice_64x30s_vs_oracle_on_hip_logprobs(oracle):
    """One rank's share of BASELINE configs[4]: 64 clips x 30 s.  Seeded random weights decode to near-empty
    transcripts, so every clip fails the 0.5 gate and takes the 0.9x / 1.1x passes.  For a sample of clips
    the three passes are recomputed by the oracle FROM THE HIP LOG-PROBS of the same (GPU-resampled) audio and
    the reference rule applied; for every clip the batched result must equal the clip run on its own."""
    from offline_tarteel_amd import plugin
    from offline_tarteel_amd.engine import Engine
    from test_tta_golden import decide

    B, N = 64, 480000
    eng = Engine(device=0, with_model=True, seed=20260630, max_batch=B, max_samples=int(N * 1.1) + 1600)
    try:
        audio = torch.from_numpy(synth_audio(B, N)).cuda()
        lens = [N - 16000 * (b % 4) for b in range(B)]
        for b, n in enumerate(lens):
            audio[b, n:] = 0
        out = plugin.tta_device_batch(eng, audio, lens)
        assert len(out) == B and all("tta" in r for r in out), "every clip must have been gated"

        def key(r):
            return (r["surah"], r["ayah"], r["ayah_end"], r.get("source"), r["tta"], [tuple(k) for k in r["tta_preds"]])

        for b in (0, 7, 22, 41, 63):
            clip = audio[b, : lens[b]].contiguous()
            variants = [eng.speed_perturb(clip, 0.9), clip, eng.speed_perturb(clip, 1.1)]
            passes = []
            for v in variants:
                lp, T = eng.forward(v[None, :].contiguous(), [int(v.numel())])
                w = oracle.predict_logprobs(lp[0, : T[0]].cpu().numpy())
                passes.append({"surah": w["surah"], "ayah": w["ayah"], "ayah_end": (w["ayah_end"] or w["ayah"]) if w["surah"] else None,
                               "score": w.get("score_raw", 0.0), "source": w["source"]} if w["surah"] else
                              {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0})
            want = decide(plugin, *passes)
            got = out[b]
            assert (got["surah"], got["ayah"], got["ayah_end"]) == (want["surah"], want["ayah"], want["ayah_end"]), b
            assert got["tta"] == want["tta"] and [tuple(k) for k in got["tta_preds"]] == [tuple(k) for k in want["tta_preds"]], b
            assert abs(got["score"] - want["score"]) <= 1e-3 * max(want["score"], 1e-3), b
            # batch invariance of the whole TTA path: the clip alone gives the same answer, bit for bit
            alone = plugin.tta_device_batch(eng, clip[None, :].contiguous(), [lens[b]])[0]
            assert key(alone) == key(got) and alone["score"] == got["score"], b
    finally:
        eng.close()


def test_tta_with_batches_in_flight_equals_one_at_a_time():
    """tta_device_batch keeps the perturbed batches in flight over the engine's contexts; the combined rows must
    be the ones the one-batch-at-a-time engine returns (seeded random weights gate every clip, so the 0.9x / 1.1x
    passes run; six clips with max_batch 4 make three perturbed batches -- more than the two contexts)."""
    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from offline_tarteel_amd.plugin import tta_device_batch

    n = 48000
    audio = torch.from_numpy(synth_audio(4, n, seed=5)).cuda()
    lens = [n, n - 3200, n - 800, n - 6400]
    for b in range(4):
        audio[b, lens[b]:] = 0
    got = {}
    for ctxs in (1, 2):
        eng = Engine(device=0, with_model=True, seed=20260630, max_batch=4, max_samples=int(n * 1.1) + 1600, contexts=ctxs)
        got[ctxs] = tta_device_batch(eng, audio, lens, want_text=True)
        torch.cuda.synchronize()
        del eng
    assert all("tta" in r for r in got[1]), "the random-weight anchors are expected to fall below the 0.5 gate"
    assert got[1] == got[2]
