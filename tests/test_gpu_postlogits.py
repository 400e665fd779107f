"""HIP post-logits stages vs the CPU oracle / reference fixtures, through the C ABI (GPU)."""

import gzip
import json

import numpy as np
import pytest
import torch

from synth import synth_logits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from offline_tarteel_amd.engine import Engine

    eng = Engine(device=0, with_model=False, max_batch=16)
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def ret_cases(golden_dir):
    return json.load(gzip.open(golden_dir / "retrieval_cases.json.gz"))


@pytest.fixture(scope="module")
def e2e_cases(golden_dir):
    return json.load(gzip.open(golden_dir / "e2e_cases.json.gz"))


def lp_of(recipe):
    lg = synth_logits(recipe["ids"], recipe["T"], seed=recipe["seed"], noise=recipe["noise"],
                      boost=recipe["boost"], rep=recipe["rep"])
    return torch.log_softmax(torch.from_numpy(lg), dim=-1)


def test_ctc_loss_kernel_vs_torch(engine, oracle):
    """float32 alpha recursion: |loss - F.ctc_loss| <= 1e-3 absolute (north_star: 1e-2)."""
    rng = np.random.default_rng(0)
    for T, noise in ((20, 1.0), (63, 2.0), (126, 1.0), (251, 3.0), (376, 1.0)):
        ids0 = oracle.token_ids(int(rng.integers(0, 6236)), 1).tolist()
        lp = torch.log_softmax(torch.from_numpy(synth_logits(ids0, T, seed=T, noise=noise, boost=5.0, rep=2)), -1)
        targets = []
        for _ in range(40):
            L = int(rng.integers(1, max(2, (T - 1) // 2 + 1)))
            t = rng.integers(0, 1024, size=L)
            if rng.random() < 0.5 and L > 2:
                t[1] = t[0]  # repeated token -> no skip transition
            targets.append(t.astype(np.uint16))
        targets.append(np.asarray(ids0[: max(1, (T - 1) // 2)], np.uint16))
        want = oracle.ctc_loss_torch(lp.numpy(), targets)
        got = engine.debug_ctc_loss(lp.cuda().contiguous(), targets)
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max() / 100), (T, np.abs(got - want).max())


def test_ctc_loss_kernel_vs_float64_twin(engine, oracle):
    """The float32 HIP recursion against the float64 restatement of the reference's browser
    rerank (lib/ctc-rescore.ts): normalised loss within 1e-5 relative (SURVEY.md 8a13 measured
    3e-7 for torch fp32 vs that recursion); both sides agree on which targets have no alignment."""
    rng = np.random.default_rng(1)
    for T, noise in ((24, 1.0), (126, 2.0), (251, 1.5)):
        ids0 = oracle.token_ids(int(rng.integers(0, 6236)), 1).tolist()
        lp = torch.log_softmax(torch.from_numpy(synth_logits(ids0, T, seed=T + 3, noise=noise, boost=5.0, rep=2)), -1)
        targets = [np.asarray(ids0[: max(1, (T - 1) // 2)], np.uint16)]
        for _ in range(12):
            L = int(rng.integers(1, max(2, (T - 1) // 2 + 1)))
            t = rng.integers(0, 1024, size=L)
            if L > 2:
                t[2] = t[1]
            targets.append(t.astype(np.uint16))
        got = engine.debug_ctc_loss(lp.cuda().contiguous(), targets)
        for tg, g in zip(targets, got):
            w = oracle.ctc_score_f64(lp.numpy(), tg.tolist())
            assert w < 1e9
            assert abs(g / len(tg) - w) <= 1e-5 * max(1.0, abs(w)), (T, len(tg), g / len(tg), w)
    # the feasibility rule of the twin is the 2L+1 <= T gate of the reference (c2c-direct/run.py:332)
    assert oracle.ctc_score_f64(lp.numpy(), list(range(126))) == 1e9
    assert oracle.ctc_score_f64(lp.numpy(), []) == 1e9


def test_parity_specialised_ctc_recursion_equals_the_generic_one_bit_for_bit(engine, oracle):
    """Round 6: ctc_wave2 (two-term log-sum-exp for the blank states of even-NS instantiations, v_max3 / v_min3 / v_med3,
    in-place update) against the wave program of rounds 1-5 -- same operands in the same order, so every loss must carry
    the same BITS, on every instantiation (1 ... 12 states per lane: targets of 1 ... 383 tokens), with repeated tokens
    (no skip transition), targets with no alignment (loss inf) and T = 2L + 1 exactly."""
    rng = np.random.default_rng(6)
    for T, noise in ((33, 1.0), (126, 2.5), (376, 1.0), (413, 3.0), (767, 2.0)):
        ids0 = oracle.token_ids(int(rng.integers(0, 6236)), 1).tolist()
        lp = torch.log_softmax(torch.from_numpy(synth_logits(ids0, T, seed=T + 11, noise=noise, boost=5.0, rep=2)), -1)
        targets = [np.asarray(ids0[: max(1, (T - 1) // 2)], np.uint16)]
        for L in sorted({1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 191, 192, 193, 255, 256, 300, 383}):
            if 2 * L + 1 > min(T, 768):
                continue
            t = rng.integers(0, 1024, size=L)
            if L > 3:
                t[2] = t[1]
                t[L - 1] = t[L - 2]
            targets.append(t.astype(np.uint16))
            targets.append(np.full(L, int(rng.integers(0, 1024)), np.uint16))      # one token repeated: needs 2L - 1 + L frames
        dev = lp.cuda().contiguous()
        try:
            engine.kernel_variant(4, 0)
            old = engine.debug_ctc_loss(dev, targets)
            engine.kernel_variant(4, 1)
            new = engine.debug_ctc_loss(dev, targets)
        finally:
            engine.kernel_variant(4, -1)
        assert old.view(np.uint32).tolist() == new.view(np.uint32).tolist(), (T, np.abs(old - new).max())
    # ... and through the hot path (leaders + the prefix read-outs of their members): corrupted recitations that fail the gate
    lps = []
    for i in range(8):
        ids = oracle.token_ids(int(rng.integers(0, 6236)), 1 + i % 3).tolist()[:60]
        lps.append(torch.log_softmax(torch.from_numpy(synth_logits(ids, 126, seed=900 + i, noise=3.5, boost=4.0, rep=2)), -1))
    dev = torch.stack(lps).cuda().contiguous()
    rows = {}
    try:
        for var in (0, 1):
            engine.kernel_variant(4, var)
            rows[var] = engine.decode_retrieve_rerank(dev, [126] * 8, want_text=False)
    finally:
        engine.kernel_variant(4, -1)
    assert sum(r["use_ctc"] for r in rows[1]) >= 4
    for a, b in zip(rows[0], rows[1]):
        assert a == b, (a, b)


def test_retrieval_matches_reference_fixtures(engine, oracle, ret_cases):
    from oracle.oracle import normalize_arabic

    checked = 0
    for c in ret_cases:
        t = c["transcript"]
        if normalize_arabic(t) != t or c["name"] == "garbage_40":
            continue  # device entry takes normalised transcripts (greedy decode output always is)
        r = engine.debug_retrieve(t)
        g = c["match"]
        key = engine.tables.key_of(r["base_start"], r["base_span"])
        want_end = g["ayah_end"] if g["ayah_end"] is not None else g["ayah"]
        assert key == (g["surah"], g["ayah"], want_end), c["name"]
        assert r["base_score"] == g["score"], c["name"]
        run = [[int(engine.tables.surah[i]), int(engine.tables.ayah[i]), round(float(s), 3)]
               for i, s in zip(r["runner_idx"], r["runner_score"])]
        assert run == g["runners_up"], c["name"]
        keys = [list(engine.tables.key_of(int(a), int(b))) for a, b in zip(r["cand_start"], r["cand_span"])]
        assert keys == c["candidates"], c["name"]
        assert r["cand_score"].tolist() == c["cand_scores"], c["name"]
        checked += 1
    assert checked >= 18


def test_retrieval_matches_oracle_on_tie_case(engine, oracle, ret_cases):
    c = [x for x in ret_cases if x["name"] == "garbage_40"][0]
    r = engine.debug_retrieve(c["transcript"])
    cs, cp, sc, m = oracle.build_candidates(c["transcript"])
    assert (r["base_start"], r["base_span"], r["base_score"]) == (m.start, m.span, m.score)
    assert r["cand_start"].tolist() == cs.tolist() and r["cand_span"].tolist() == cp.tolist()
    assert r["cand_score"].tolist() == sc.tolist()


def test_end_to_end_fixtures_batched(engine, oracle, e2e_cases):
    """all e2e fixtures in ONE ragged batch: greedy ids, transcript, winner, score."""
    lps = [lp_of(c["recipe"]) for c in e2e_cases]
    t_max = max(x.shape[0] for x in lps)
    B = len(lps)
    batch = torch.full((B, t_max, 1025), -50.0)
    for b, x in enumerate(lps):
        batch[b, : x.shape[0]] = x
    res = engine.decode_retrieve_rerank(batch.cuda().contiguous(), [x.shape[0] for x in lps])
    for c, r in zip(e2e_cases, res):
        g = c["result"]
        assert r["greedy_ids"] == c["greedy_ids"], c["name"]
        assert r["transcript"] == c["transcript"], c["name"]
        assert (r["surah"], r["ayah"], r["ayah_end"], r["source"]) == (
            g["surah"], g["ayah"], g["ayah_end"], g["source"]), c["name"]
        if g["source"] == "text":
            assert r["score"] == g["score_raw"], c["name"]
        elif g["source"] == "ctc":
            assert abs(r["score"] - g["score_raw"]) <= 1e-3 * max(g["score_raw"], 1e-3), c["name"]
            assert round(r["score"], 4) == g["score"] or abs(r["score"] - g["score_raw"]) < 1e-6, c["name"]
        if "use_ctc" in c:
            assert r["use_ctc"] == c["use_ctc"], c["name"]
            if c["use_ctc"]:
                assert r["n_candidates"] == c["n_candidates"], c["name"]


def test_per_candidate_ctc_losses_of_the_fixtures(engine, e2e_cases):
    """The reference's whole per-candidate `ctc_loss` vector (F.ctc_loss, float32) and its ranking, for every
    e2e fixture: the candidates' token ids (table lookups of the fixture's (surah, ayah, ayah_end) keys) go
    through the rerank kernel's recursion (qv_debug_ctc_loss) on the fixture's log-probs.  Infeasible
    candidates (2L+1 > T) are the ones the fixture records as null."""
    tb = engine.tables
    checked = 0
    for c in e2e_cases:
        if "cand_keys" not in c or not c["cand_keys"]:
            continue
        lp = lp_of(c["recipe"])
        T = lp.shape[0]
        starts = [tb.verse_index(s, a) for s, a, _ in c["cand_keys"]]
        spans = [e - a + 1 for _, a, e in c["cand_keys"]]
        ids = [tb.token_ids(st, sp) for st, sp in zip(starts, spans)]
        feas = [i for i, x in enumerate(ids) if len(x) > 0 and 2 * len(x) + 1 <= T]
        assert feas == [i for i, w in enumerate(c["ctc_loss"]) if w is not None], c["name"]
        # the reference records the target length of the candidates it scored (0 for the gated-out ones)
        assert [len(ids[i]) if i in set(feas) else 0 for i in range(len(ids))] == c["ctc_len"], c["name"]
        if not feas:
            continue
        got = engine.debug_ctc_loss(lp.cuda().contiguous(), [ids[i] for i in feas])
        want = np.array([c["ctc_loss"][i] for i in feas], np.float64)
        assert np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max() / 100), (c["name"], np.abs(got - want).max())
        # ranking: final = -loss / len - 0.5 * (span - 1), stable in candidate order (c2c-direct/run.py:366-379)
        final = [-(float(g) / len(ids[i])) - 0.5 * (spans[i] - 1) for g, i in zip(got, feas)]
        order = sorted(range(len(feas)), key=lambda k: -final[k])
        top = [c["cand_keys"][feas[k]] for k in order[:20]]
        gaps_ok = all(abs(a - b) > 2e-3 for a, b in zip(c["ranked_final"][:-1], c["ranked_final"][1:]))
        if gaps_ok:
            assert top == c["ranked_keys"], c["name"]
        else:
            assert top[0] == c["ranked_keys"][0] and sorted(map(tuple, top)) == sorted(map(tuple, c["ranked_keys"])), c["name"]
        assert np.allclose([final[k] for k in order[:20]], c["ranked_final"], atol=2e-3), c["name"]
        checked += 1
    assert checked >= 6


def test_long_transcripts_T376_vs_oracle(oracle):
    """30 s worth of frames: long verse prefixes (transcripts of several hundred characters ->
    multi-word bit-vectors, windows on both sides, CTC targets with > 256 states)."""
    import random

    from offline_tarteel_amd.engine import Engine

    eng = Engine(device=0, with_model=False, max_batch=4, max_samples=480000)
    rnd = random.Random(5)
    lps, want = [], []
    for (s, a), keep, rate in (((2, 282), 170, 0.0), ((2, 282), 150, 0.25), ((4, 12), 120, 0.1), ((2, 255), 80, 0.4)):
        ids = oracle.token_ids(oracle.verse_index(s, a), 1).tolist()[:keep]
        ids = [(rnd.randrange(1, 1024) if rnd.random() < rate else i) for i in ids]
        lg = synth_logits(ids, 376, seed=s * 1000 + a, noise=1.0, boost=7.0, rep=2)
        lp = torch.log_softmax(torch.from_numpy(lg), -1)
        lps.append(lp)
        want.append(oracle.predict_logprobs(lp.numpy()))
    res = eng.decode_retrieve_rerank(torch.stack(lps).cuda().contiguous(), [376] * len(lps))
    for got, w in zip(res, want):
        assert got["greedy_ids"] == w["greedy_ids"]
        assert got["transcript"] == w["transcript"] and len(w["transcript"]) > 150
        assert (got["surah"], got["ayah"], got["ayah_end"], got["source"]) == (
            w["surah"], w["ayah"], w["ayah_end"], w["source"])
        assert got["use_ctc"] == w.get("use_ctc", got["use_ctc"])
        if got["use_ctc"]:
            assert got["n_candidates"] == w["n_candidates"]
        assert abs(got["score"] - w.get("score_raw", 0.0)) <= 1e-3 * max(w.get("score_raw", 0.0), 1e-3)
    eng.close()


def test_random_transcripts_vs_oracle(engine, oracle):
    """seeded perturbed verses / spans: device == oracle for base + full candidate list."""
    import random

    from oracle.oracle import normalize_arabic

    rnd = random.Random(11)
    letters = [ch for ch in oracle.alphabet if ch != " "][:28]
    for it in range(12):
        v = rnd.randrange(6236)
        span = rnd.choice([1, 1, 2, 3])
        last = int(oracle.t["surah_start"][oracle.surah[v]]) - 1
        span = max(1, min(span, last - v + 1))
        text = " ".join(oracle.verse_text(v + k) for k in range(span))
        rate = rnd.choice([0.0, 0.1, 0.3, 0.6])
        out = []
        for ch in text:
            x = rnd.random()
            if x < rate / 3:
                continue
            out.append(rnd.choice(letters) if x < 2 * rate / 3 else ch)
        # the device entry takes normalised transcripts (what greedy decode always produces)
        t = normalize_arabic(" ".join("".join(out).split())[:900])
        if not t:
            continue
        r = engine.debug_retrieve(t)
        cs, cp, sc, m = oracle.build_candidates(t)
        assert (r["base_start"], r["base_span"], r["base_score"]) == (m.start, m.span, m.score), (it, t)
        assert r["cand_start"].tolist() == cs.tolist(), it
        assert r["cand_span"].tolist() == cp.tolist(), it
        assert r["cand_score"].tolist() == sc.tolist(), it


def test_span_pass_with_the_pattern_spread_over_lanes(engine, oracle):
    """k_spans spreads the words of a long transcript over 4 / 8 / 16 neighbouring lanes (lcs_systolic: skewed wavefront,
    carry by DPP row shift).  Corrupted multi-ayah recitations of ~150 ... ~1,000 normalised characters (3 ... 16 words of
    64 pattern bits: every group size, both ends of each) against the oracle: the winner of match_verse -- usually a span --
    its score bit for bit, and the whole candidate list in order."""
    import random

    from oracle.oracle import normalize_arabic

    rnd = random.Random(404)
    letters = [ch for ch in oracle.alphabet if ch != " "][:28]
    seen_w = set()
    for target in (150, 200, 250, 260, 330, 500, 515, 640, 770, 900, 1000, 1024):
        for rate in (0.15, 0.45):
            v = rnd.randrange(210, 280) if target >= 640 else rnd.randrange(6000)   # (the long ayat of surah 2 for the long ones)
            words = []
            k = 0
            while sum(len(w) + 1 for w in words) < target + 80 and k < 8:      # up to 8 consecutive ayat
                last = int(oracle.t["surah_start"][oracle.surah[v]]) - 1
                if v + k > last:
                    break
                words += oracle.verse_text(v + k).split()
                k += 1
            out = []
            for ch in " ".join(words):
                x = rnd.random()
                if x < rate / 3:
                    continue
                out.append(rnd.choice(letters) if x < 2 * rate / 3 else ch)
            t = normalize_arabic(" ".join("".join(out).split())[:target]).strip()
            if len(t) < 130:
                continue
            seen_w.add((len(t) + 63) // 64)
            r = engine.debug_retrieve(t)
            cs, cp, sc, m = oracle.build_candidates(t)
            assert (r["base_start"], r["base_span"], r["base_score"]) == (m.start, m.span, m.score), (target, rate, len(t))
            assert r["cand_start"].tolist() == cs.tolist() and r["cand_span"].tolist() == cp.tolist(), (target, rate)
            assert r["cand_score"].tolist() == sc.tolist(), (target, rate)
    assert {3, 4, 5, 8, 9, 16} <= seen_w, seen_w      # both ends of G = 4, 8 and 16


def test_prefix_shared_span_pass_equals_one_walk_per_span(engine, oracle):
    """k_spans2 (one walk per START verse over the 8-code-padded text, LCS read off at every ayah end: the default) against
    k_spans (one walk per span): the same spans survive the same exact bound and score the same integers, so base match,
    candidate list and scores are identical -- short transcripts (one start verse per lane) and long ones (4 / 8 / 16 lanes
    per start verse, chunk-skewed), spans that start with a bismillah-less first verse included."""
    import random

    from oracle.oracle import normalize_arabic

    rnd = random.Random(77)
    texts = []
    for v0, k in ((0, 3), (7, 4), (1, 6), (293, 2), (6225, 5), (6230, 6), (255, 3), (2000, 4), (4000, 6), (5000, 2)):
        words = []
        for d in range(k):
            words += oracle.verse_text(v0 + d).split()
        full = " ".join(words)
        for cut in (40, 100, 127, 128, 129, 200, 400, 900):
            out = [ch for ch in full[:cut] if rnd.random() > 0.06]
            t = normalize_arabic(" ".join("".join(out).split())).strip()
            if len(t) >= 12:
                texts.append(t)
    assert len({(len(t) + 63) // 64 for t in texts}) >= 5
    try:
        got = {}
        for var in (0, 1):
            engine.kernel_variant(2, var)
            got[var] = [engine.debug_retrieve(t) for t in texts]
    finally:
        engine.kernel_variant(2, -1)
    for t, a, c in zip(texts, got[0], got[1]):
        assert (a["base_start"], a["base_span"], a["base_score"]) == (c["base_start"], c["base_span"], c["base_score"]), t
        assert a["cand_start"].tolist() == c["cand_start"].tolist() and a["cand_span"].tolist() == c["cand_span"].tolist(), t
        assert a["cand_score"].tolist() == c["cand_score"].tolist(), t
    # and against the oracle for a few of them
    for t, c in list(zip(texts, got[1]))[::7]:
        cs, cp, sc, m = oracle.build_candidates(t)
        assert (c["base_start"], c["base_span"], c["base_score"]) == (m.start, m.span, m.score), t


def test_prefix_shared_span_pass_under_match_verse_with_spans_of_eight(engine, golden_dir):
    """qv_match_verse (no trigram restriction, max_span = 8) on long low-confidence transcripts whose candidate starts sit in
    runs of very short ayat (surah 74: two 8-code chunks per ayah): the skewed lanes of k_spans2 step past the end of the
    walk and must not read further ayah ends into the snapshot fields (a first version did: 74:31-32 at score 1.0)."""
    import json

    cases = json.loads((golden_dir / "longtx_cases.json").read_text(encoding="utf-8"))
    words = cases[1]["text"].split()
    texts = [" ".join(words[:k]) for k in (40, 60, 90, 120, 150, 200, 259, len(words))] + [" ".join(cases[0]["text"].split()[:k]) for k in (50, 130, 220)]
    try:
        got = {}
        for var in (0, 1):
            engine.kernel_variant(2, var)
            got[var] = [engine.match_verse(t, threshold=0.0, max_span=8) for t in texts]
    finally:
        engine.kernel_variant(2, -1)
    assert got[0] == got[1]
    assert all(g is not None and g["score"] < 1.0 for g in got[1][:8])


def test_span_pass_differential_fuzz_small():
    """tools/fuzz_spans.py at a size that runs in seconds (the full runs: profiles/archive/r05_l_fuzz_spans.log, 33,000 texts, 0
    mismatches): k_spans2 against k_spans through the hot path's retrieval and through qv_match_verse (max_span 8, hints)."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "tools" / "fuzz_spans.py"), "--n", "400", "--seed", "9"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "400 texts, 0 mismatches" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_text_weight_fixtures(golden_dir):
    """CTC_DIRECT_TEXT_WEIGHT != 0 on the device: the reference's winner (and exp(-norm_loss) score) for weights 0.35 and
    2.0 -- one of the three recipes changes its winner between the two."""
    from offline_tarteel_amd.engine import Engine

    cases = json.load(gzip.open(golden_dir / "e2e_textweight_cases.json.gz"))
    for tw in (0.35, 2.0):
        eng = Engine(device=0, with_model=False, max_batch=4, max_samples=200000, text_weight=tw)
        try:
            sel = [c for c in cases if c["text_weight"] == tw]
            lps = [lp_of(c["recipe"]) for c in sel]
            t_max = max(x.shape[0] for x in lps)
            batch = torch.full((len(lps), t_max, 1025), -50.0)
            for b, x in enumerate(lps):
                batch[b, : x.shape[0]] = x
            res = eng.decode_retrieve_rerank(batch.cuda().contiguous(), [x.shape[0] for x in lps])
            for c, r in zip(sel, res):
                assert r["transcript"] == c["transcript"] and r["use_ctc"] and r["n_candidates"] == c["n_candidates"], c["name"]
                assert [r["surah"], r["ayah"], r["ayah_end"]] == c["winner"] and r["source"] == "ctc", (c["name"], tw, r)
                assert abs(r["score"] - c["winner_score_raw"]) <= 1e-3 * max(c["winner_score_raw"], 1e-3), (c["name"], tw)
        finally:
            eng.close()


def _fuzz_recipe(rng, oracle):
    """token ids of a random verse / span, corrupted in one of several ways, plus the frame count and noise level."""
    n_verses = 6236
    v = int(rng.integers(0, n_verses))
    span = int(rng.choice([1, 1, 1, 2, 3]))
    ids = oracle.token_ids(v, span).tolist() if v + span <= n_verses else oracle.token_ids(v, 1).tolist()
    mode = int(rng.integers(0, 9))
    if mode == 1:                                   # tokens dropped
        keep = rng.random(len(ids)) > rng.uniform(0.1, 0.5)
        ids = [t for t, k in zip(ids, keep) if k]
    elif mode == 2:                                 # tokens replaced
        rate = rng.uniform(0.1, 0.4)
        ids = [int(rng.integers(1, 1024)) if rng.random() < rate else t for t in ids]
    elif mode == 3:                                 # random insertions
        out = []
        for t in ids:
            out.append(t)
            if rng.random() < 0.2:
                out.append(int(rng.integers(1, 1024)))
        ids = out
    elif mode == 4:                                 # a middle fragment
        a = int(rng.integers(0, max(1, len(ids) // 2)))
        ids = ids[a: a + max(2, len(ids) // 2)]
    elif mode == 5:                                 # two unrelated verses glued together
        ids = ids[: max(2, len(ids) // 2)] + oracle.token_ids(int(rng.integers(0, n_verses)), 1).tolist()[:40]
    elif mode == 6:                                 # garbage
        ids = [int(x) for x in rng.integers(1, 1024, size=int(rng.integers(3, 60)))]
    elif mode == 7:                                 # very short
        ids = ids[: int(rng.integers(1, 4))]
    ids = ids[:180] or [5]
    T = int(min(376, max(12, round(len(ids) * rng.uniform(2.05, 2.6)) + 1)))
    return {"ids": ids, "T": T, "seed": int(rng.integers(0, 1 << 30)), "noise": float(rng.choice([0.5, 1.0, 2.0, 3.0])),
            "boost": float(rng.choice([4.0, 6.0, 8.0])), "rep": 2}


def test_fuzz_batched_path_against_the_oracle(oracle):
    """Random corrupted recitations through the BATCHED path (ragged batches of 32: the fragment job list, the work
    stealing and the leader assignment see many utterances at once) against the CPU oracle one utterance at a time:
    greedy ids, winner, source, candidate count, score.  QVERSE_FUZZ_CASES scales it up (default 128)."""
    import os

    from offline_tarteel_amd.engine import Engine

    n_cases = int(os.getenv("QVERSE_FUZZ_CASES", "128"))
    rng = np.random.default_rng(int(os.getenv("QVERSE_FUZZ_SEED", "20260927")))
    eng = Engine(device=0, with_model=False, max_batch=32, max_samples=480000)
    try:
        done = withheld = 0
        while done < n_cases:
            recipes = [_fuzz_recipe(rng, oracle) for _ in range(min(32, n_cases - done))]
            lps = [lp_of(r) for r in recipes]
            t_max = max(x.shape[0] for x in lps)
            batch = torch.full((len(lps), t_max, 1025), -50.0)
            for b, x in enumerate(lps):
                batch[b, : x.shape[0]] = x
            res = eng.decode_retrieve_rerank(batch.cuda().contiguous(), [x.shape[0] for x in lps])
            for rcp, lp, got in zip(recipes, lps, res):
                want = oracle.predict_logprobs(lp.numpy())
                tag = (rcp["seed"], len(rcp["ids"]), rcp["T"])
                assert got["greedy_ids"] == want["greedy_ids"], tag
                if len(want["transcript"]) > 1024:
                    # documented deviation: a transcript of more than 1,024 normalised characters -- only
                    # noise decodes to one -- is withheld on the hot path; the reference would still match it
                    assert got["surah"] == 0, tag
                    withheld += 1
                    continue
                assert (got["surah"], got["ayah"], got["ayah_end"], got["source"]) == (
                    want["surah"], want["ayah"], want["ayah_end"], want["source"]), (tag, got, want)
                if want["source"] is None:
                    continue
                assert got["use_ctc"] == want["use_ctc"], tag
                if want["use_ctc"]:
                    assert got["n_candidates"] == want["n_candidates"], tag
                if want["source"] == "text":
                    assert got["score"] == want["score_raw"], tag
                else:
                    assert abs(got["score"] - want["score_raw"]) <= 1e-3 * max(want["score_raw"], 1e-3), tag
            done += len(recipes)
        assert withheld <= n_cases // 50
    finally:
        eng.close()
