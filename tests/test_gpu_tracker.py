"""Streaming row on the MI355X: qv_tracker_match (the HIP matching step of the verse tracker)
against the reference-generated golden fixtures and the CPU oracle, bit-exact in the fp64 scores,
and the streaming pipeline end to end through the C ABI."""

import gzip
import json
import random

import numpy as np
import pytest

from synth import synth_audio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from offline_tarteel_amd.engine import Engine

    eng = Engine(device=0, with_model=False, max_batch=16)
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def cases(golden_dir):
    with gzip.open(golden_dir / "tracker_cases.json.gz", "rt", encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tracker_oracle(oracle):
    from oracle.tracker_ref import TrackerOracle

    return TrackerOracle(oracle)


def gate(m, text, streaming):
    """the caller-side gates of _find_best_match (verse_tracker.py:72-76, 98-100)"""
    if not text.strip() or (streaming and len(text.split()) < 2) or m is None:
        return None
    return m if m["score"] >= (0.4 if streaming else 0.3) else None


def test_track_match_golden(engine, cases):
    """All 356 reference cases in ONE call (more than QV_TRACK_CAP texts: exercises the slicing)."""
    cs = cases["best_match"]
    got = engine.track_match([c["text"] for c in cs], [tuple(c["last"]) if c["last"] else None for c in cs])
    assert len(got) == len(cs)
    for c, g in zip(cs, got):
        m, w = gate(g, c["text"], c["streaming"]), c["match"]
        assert (m is None) == (w is None), c["text"]
        if m:
            assert (m["surah"], m["ayah"], m["n_words"]) == (w["surah"], w["ayah"], w["n_words"]), c["text"]
            assert m["score"] == w["score"], (c["text"], m["score"], w["score"])


def test_track_match_vs_oracle_random(engine, tracker_oracle):
    """Seeded corrupted / truncated / glued verse texts, with and without a continuation verse;
    verse index, matched variant, word count and fp64 score must equal the oracle's."""
    rng = random.Random(7)
    o = tracker_oracle.o
    alphabet = [ch for ch in o.alphabet if ch != " "]
    texts, lasts = [], []
    for k in range(80):
        v = rng.randrange(6236 - 2)
        t = " ".join(o.verse_text(v + j) for j in range(rng.choice((1, 1, 2, 3))))
        w = t.split()
        if rng.random() < 0.5:
            w = w[: max(1, rng.randrange(len(w) + 1))]
        chars = list(" ".join(w))
        for i in range(len(chars)):
            if chars[i] != " " and rng.random() < 0.12:
                chars[i] = rng.choice(alphabet + ["x"])          # "x": outside the verse alphabet
        t = " ".join("".join(chars).split())[:1000]
        texts.append(t)
        s, a = int(o.surah[v]), int(o.ayah[v])
        lasts.append(rng.choice([None, (s, a - 1) if a > 1 else None, (s, a), (114, 6), (3, 999)]))
    texts += ["", "x y z"]
    lasts += [None, None]
    got = engine.track_match(texts, lasts)
    for t, last, g in zip(texts, lasts, got):
        w = tracker_oracle.best_raw(t, last) if t else None
        assert (g is None) == (w is None), t
        if g:
            assert (g["verse"], g["variant"], g["n_words"]) == (w[0], w[1], w[2]), t
            assert g["score"] == w[3], t


def test_track_match_capacity(engine, tracker_oracle):
    """Texts longer than the device's 1,024-character window are matched on their longest whole-word front
    window instead of being refused (the reference has no limit; both of its single-text matchers work on
    the FRONT of the text): the Python binding windows them, the raw C entry point still reports capacity."""
    import ctypes as C

    import numpy as np

    from offline_tarteel_amd.engine import QvTrackMatch, front_window

    assert engine.track_match(["ا" * 1024])[0] is not None       # QV_MAX_TRANSCRIPT codes still fit
    words = " ".join(["الحمد لله رب العالمين"] * 60)             # 1,379 characters
    assert len(words) > 1024
    win = front_window(words)
    assert len(win) <= 1024 and words.startswith(win) and words[len(win)] == " "
    got, want = engine.track_match([words])[0], engine.track_match([win])[0]
    assert got == want and got is not None
    w = tracker_oracle.best_raw(win, None)
    assert (got["verse"], got["variant"], got["n_words"], got["score"]) == w
    assert engine.match_verse(words) == engine.match_verse(win)
    # the C ABI itself refuses (QV_ERR_CAPACITY = 4) rather than truncating silently
    codes = np.zeros(1025, np.uint8)
    off, nw, bonus = np.array([0, 1025], np.int32), np.array([1], np.int32), np.array([-1], np.int32)
    out = (QvTrackMatch * 1)()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert engine.lib.qv_tracker_match(engine.h, p(codes), p(off), p(nw), p(bonus), 1, C.cast(out, C.c_void_p), None) == 4
    assert engine.track_match([]) == []


def test_pipeline_run_on_text_golden(engine, cases):
    from offline_tarteel_amd.streaming import StreamingPipeline

    pipe = StreamingPipeline(engine)
    for c in cases["run_on_text"]:
        assert pipe.run_on_text(c["snapshots"]) == c["emissions"]


def test_pipeline_chunked_golden(engine, cases):
    """Scripted transcribe_fn (str and {"text","avg_logprob"} returns) over the reference's chunk
    walk: emissions equal the reference's, one recording at a time and all recordings in lock step
    (one qv_tracker_match launch per round for all of them)."""
    from offline_tarteel_amd.streaming import StreamingPipeline

    pipe = StreamingPipeline(engine)
    for c in cases["chunked"]:
        calls = []

        def fn(path, c=c, calls=calls):
            i = len(calls)
            calls.append(path)
            return c["script"][i] if i < len(c["script"]) else ""

        got = pipe.run_on_audio_chunked(np.zeros(c["n_samples"], np.float32), fn, chunk_seconds=c["chunk_seconds"],
                                        overlap_seconds=c["overlap_seconds"])
        assert len(calls) == c["n_calls"] and got == c["emissions"]
    # lock step: same chunking for a group of recordings, scripts looked up by temp-file call order
    group = [c for c in cases["chunked"] if c["chunk_seconds"] == 3.0 and c["overlap_seconds"] == 0.0]
    assert len(group) >= 2
    from offline_tarteel_amd import streaming as st

    class Scripted(StreamingPipeline):
        def transcribe_chunks(self, chunk_lists):
            return [[(c["script"][k] if k < len(c["script"]) else "") for k in range(len(chunks))]
                    for c, chunks in zip(group, chunk_lists)]

    got = Scripted(engine).run_on_audio_chunked_batch([np.zeros(c["n_samples"], np.float32) for c in group])
    assert got == [c["emissions"] for c in group]
    assert st.MAX_HOLD_CHUNKS == 3


def test_pipeline_audio_end_to_end():
    """Engine as the ASR backend (seeded synthetic weights): every chunk of every recording goes
    through one packed ragged forward; the batch result equals the one-recording-at-a-time result,
    and the chunk transcripts equal transcribing each chunk alone."""
    import torch

    from offline_tarteel_amd.engine import Engine
    from offline_tarteel_amd.streaming import StreamingPipeline, split_chunks

    eng = Engine(device=0, with_model=True, seed=7, max_batch=8, max_samples=16000 * 4)
    try:
        pipe = StreamingPipeline(eng)
        audio = synth_audio(3, 16000 * 8)
        recs = [audio[0], audio[1][: 16000 * 5 + 9000], audio[2][: 16000 * 3 + 4000]]
        chunk_lists = [split_chunks(r) for r in recs]
        assert [len(c) for c in chunk_lists] == [3, 2, 1]
        texts = pipe.transcribe_chunks(chunk_lists)
        for chunks, tx in zip(chunk_lists, texts):
            for c, t in zip(chunks, tx):
                alone = eng.transcribe_batch(torch.from_numpy(c[None, :]).cuda(), [len(c)])[0]
                assert alone == t
        together = pipe.run_on_audio_chunked_batch(recs)
        assert together == [pipe.run_on_audio_chunked(r) for r in recs]
    finally:
        eng.close()


def test_reference_unit_scenarios(engine, oracle):
    """The scenarios of the reference's own unit tests for this row (tests/test_verse_tracker.py,
    tests/test_streaming_pipeline.py, tests/test_quran_db.py), with the HIP matching step."""
    from offline_tarteel_amd.streaming import StreamingPipeline
    from offline_tarteel_amd.verse_tracker import VerseTracker

    def verse(s, a):
        return oracle.verse_text(oracle.verse_index(s, a))

    def run_text(text, **kw):
        tr = VerseTracker(engine, **kw)
        return tr.process_text(text) + tr.finalize()

    e = run_text(verse(1, 1))
    assert (e[0]["surah"], e[0]["ayah"]) == (1, 1)
    e = run_text(verse(112, 1) + " " + verse(112, 2))
    assert [(x["surah"], x["ayah"]) for x in e[:2]] == [(112, 1), (112, 2)]
    e = run_text(verse(55, 13), last_emission=(55, 12))            # the refrain: continuation decides
    assert (e[0]["surah"], e[0]["ayah"]) == (55, 13)
    e = run_text("الله لا اله الا هو الحي القيوم لا تاخذه سنه ولا نوم")   # enough of 2:255 to beat 3:2
    assert (e[0]["surah"], e[0]["ayah"]) == (2, 255)
    assert run_text("") == []
    tr = VerseTracker(engine, streaming_mode=True)
    assert tr.process_delta("يا") + tr.finalize() == []             # one word: below MIN_WORDS_FOR_MATCH
    tr = VerseTracker(engine, streaming_mode=True)
    assert tr.process_delta("بسم") == []
    e = tr.process_delta(verse(1, 1)) + tr.finalize()
    assert e and e[0]["surah"] == 1

    pipe = StreamingPipeline(engine)
    e = pipe.run_on_text([verse(1, 1)])
    assert (e[0]["surah"], e[0]["ayah"]) == (1, 1)
    v1, v2, v3 = verse(103, 1), verse(103, 2), verse(103, 3)
    e = pipe.run_on_text([v1, v1 + " " + v2, v1 + " " + v2 + " " + v3])
    assert [(x["surah"], x["ayah"]) for x in e[:3]] == [(103, 1), (103, 2), (103, 3)]

    calls = []

    def low_then_good(path):
        calls.append(path)
        if len(calls) == 1:
            return {"text": "يا المسلمون الكرام", "avg_logprob": -2.0}   # gated: low confidence
        return {"text": verse(112, 1), "avg_logprob": -0.3}

    e = pipe.run_on_audio_chunked(np.zeros(16000 * 6, np.float32), low_then_good, chunk_seconds=3.0)
    assert len(calls) == 2 and (e[0]["surah"], e[0]["ayah"]) == (112, 1)
    e = pipe.run_on_audio_chunked(np.zeros(16000 * 3, np.float32), lambda p: verse(112, 1), chunk_seconds=3.0)
    assert e and e[0]["surah"] == 112
    e = pipe.run_on_audio_chunked(np.zeros(16000 * 3, np.float32),
                                  lambda p: {"text": verse(112, 1), "avg_logprob": -0.3}, chunk_seconds=3.0)
    assert e and (e[0]["surah"], e[0]["ayah"]) == (112, 1) and e[0]["score"] >= 0.7
    e = pipe.run_on_full_transcript("x.wav", lambda p: verse(112, 1))     # mock transcribe, like the reference's test
    assert e and (e[0]["surah"], e[0]["ayah"]) == (112, 1)

    # QuranDB.get_next_verse navigation (tests/test_quran_db.py)
    assert engine.next_verse(1, 1) == oracle.verse_index(1, 2)
    assert engine.next_verse(1, 7) == oracle.verse_index(2, 1)
    assert engine.next_verse(114, 6) == -1
    assert engine.next_verse(999, 1) == -1


def test_match_verse_hint_golden_and_oracle(engine, golden_dir, oracle):
    """qv_match_verse (full scan + continuation hint + spans up to 8) against the reference-generated
    fixtures and, on seeded texts, against the oracle: same winner, bit-identical fp64 score."""
    from oracle.tracker_ref import MatchVerseOracle

    with gzip.open(golden_dir / "fulltx_cases.json.gz", "rt", encoding="utf-8") as f:
        fx = json.load(f)
    for c in fx["match"]:
        r = engine.match_verse(c["text"], max_span=8, hint=tuple(c["hint"]) if c["hint"] else None)
        w = c["result"]
        assert (r is None) == (w is None), c["text"]
        if r:
            for k in ("surah", "ayah", "ayah_end", "score", "n_words"):
                assert r[k] == w[k], (c["text"], k, r[k], w[k])
    mv = MatchVerseOracle(oracle)
    rng = random.Random(11)
    for _ in range(12):
        v = rng.randrange(6236 - 4)
        k = rng.choice((1, 2, 3))
        words = " ".join(oracle.verse_text(v + j) for j in range(k)).split()
        lo = rng.randrange(0, max(1, len(words) // 3))
        text = " ".join(words[lo: lo + rng.randrange(3, 40)])[:900]
        s, a = int(oracle.surah[v]), int(oracle.ayah[v])
        hint = rng.choice([None, (s, a - 1) if a > 1 else None, (s, a), (s - 1, 999) if s > 1 else None])
        for max_span in (3, 8):
            r = engine.match_verse(text, threshold=0.0, max_span=max_span, hint=hint)
            w = mv.match_verse(text, threshold=0.0, max_span=max_span, hint=hint)
            assert (r["verse"], r["span"], r["score"], r["n_words"]) == (w["verse"], w["span"], w["score"], w["n_words"]), \
                (text, hint, max_span, r, w)
    from offline_tarteel_amd.engine import QvError

    with pytest.raises(QvError):
        engine.match_verse("قل هو الله احد", max_span=9)


def test_run_on_full_transcript_golden(engine, golden_dir):
    from offline_tarteel_amd.streaming import StreamingPipeline

    with gzip.open(golden_dir / "fulltx_cases.json.gz", "rt", encoding="utf-8") as f:
        fx = json.load(f)
    pipe = StreamingPipeline(engine)
    for c in fx["full"]:
        assert pipe.run_on_full_transcript("x.wav", lambda p, t=c["text"]: t) == c["emissions"], c["text"]


def test_run_on_full_transcript_longer_than_the_device_window(engine, golden_dir):
    """Transcripts of more than 1,024 characters (tests/golden/gen_longtx_golden.py: the reference's own
    run_on_full_transcript on 1,378 and 1,556 characters).  The device matches the longest whole-word front
    window instead of the whole text -- a documented difference -- so the peel loop neither raises nor stalls;
    on the five-ayah recitation of 2:282-286 it still emits exactly the reference's verse sequence."""
    import json

    from offline_tarteel_amd.streaming import StreamingPipeline

    cases = json.loads((golden_dir / "longtx_cases.json").read_text(encoding="utf-8"))
    assert all(c["chars"] > 1024 for c in cases)
    pipe = StreamingPipeline(engine)
    got = [pipe.run_on_full_transcript("x.wav", lambda p, t=c["text"]: t) for c in cases]
    keys = lambda em: [(e["surah"], e["ayah"]) for e in em]  # noqa: E731
    assert keys(got[0]) == keys(cases[0]["emissions"]) == [(2, a) for a in range(282, 287)]
    # 42 short ayat of surah 26 in one transcript: no window of 8 ayat comes close to 1,556 characters, the
    # reference itself answers 2:244-251 at 0.478; the front window answers another long low-confidence span
    # (5:110-117 at 0.481).  Neither is right; what is pinned is that the call completes with a bounded list.
    assert cases[1]["emissions"][0]["score"] < 0.5
    assert isinstance(got[1], list) and len(got[1]) <= 20 * 8 and all(e["score"] < 0.6 for e in got[1][:8])
