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


def test_track_match_capacity(engine):
    from offline_tarteel_amd.engine import QvError

    assert engine.track_match(["ا" * 1024])[0] is not None       # QV_MAX_TRANSCRIPT codes still fit
    with pytest.raises(QvError):
        engine.track_match(["ا" * 1025])
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
