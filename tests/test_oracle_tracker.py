"""Streaming row on the CPU: the tracker oracle against the golden fixtures generated from the
reference's VerseTracker / StreamingPipeline (tests/golden/gen_tracker_golden.py), and the
product's HOST logic (offline-tarteel_amd/verse_tracker.py, streaming.py) against the golden
emission lists with the oracle injected as the matching step.  The product's own matching step is
the HIP kernel and is checked in tests/test_gpu_tracker.py."""

import gzip
import json

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cases(golden_dir):
    with gzip.open(golden_dir / "tracker_cases.json.gz", "rt", encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tracker_oracle(oracle):
    from oracle.tracker_ref import TrackerOracle

    return TrackerOracle(oracle)


def oracle_matcher(tr):
    def fn(texts, last_refs):
        out = []
        for t, last in zip(texts, last_refs):
            b = tr.best_raw(t, last)
            out.append(None if b is None else
                       {"surah": int(tr.o.surah[b[0]]), "ayah": int(tr.o.ayah[b[0]]), "n_words": b[2], "score": b[3],
                        "verse": b[0], "variant": b[1]})
        return out
    return fn


def test_oracle_best_match_matches_reference(cases, tracker_oracle):
    # every second case keeps the CPU suite short; the GPU suite checks all of them
    for c in cases["best_match"][::2]:
        m = tracker_oracle.find_best_match(c["text"], tuple(c["last"]) if c["last"] else None, c["streaming"])
        w = c["match"]
        assert (m is None) == (w is None), c["text"]
        if m:
            assert (m["surah"], m["ayah"], m["n_words"]) == (w["surah"], w["ayah"], w["n_words"]), c["text"]
            assert m["score"] == w["score"], c["text"]   # bit-exact double arithmetic


def test_next_verse(tracker_oracle):
    assert tracker_oracle.next_verse((1, 7)) == 7          # 2:1
    assert tracker_oracle.next_verse((114, 6)) is None
    assert tracker_oracle.next_verse((1, 8)) is None       # no such ayah
    assert tracker_oracle.next_verse(None) is None


def test_host_run_on_text_matches_reference(cases, tracker_oracle):
    from offline_tarteel_amd.streaming import StreamingPipeline

    pipe = StreamingPipeline(matcher=oracle_matcher(tracker_oracle))
    picked = [c for c in cases["run_on_text"] if len(c["snapshots"]) <= 12][:6]
    assert picked
    for c in picked:
        assert pipe.run_on_text(c["snapshots"]) == c["emissions"]


def test_host_chunked_matches_reference(cases, tracker_oracle):
    from offline_tarteel_amd.streaming import StreamingPipeline, split_chunks

    pipe = StreamingPipeline(matcher=oracle_matcher(tracker_oracle))
    picked = sorted(cases["chunked"], key=lambda c: c["n_calls"])[:8]
    for c in picked:
        audio = np.zeros(c["n_samples"], np.float32)
        assert len(split_chunks(audio, c["chunk_seconds"], c["overlap_seconds"])) == c["n_calls"]
        calls = []

        def fn(path, c=c, calls=calls):
            i = len(calls)
            calls.append(path)
            return c["script"][i] if i < len(c["script"]) else ""

        got = pipe.run_on_audio_chunked(audio, fn, chunk_seconds=c["chunk_seconds"], overlap_seconds=c["overlap_seconds"])
        assert len(calls) == c["n_calls"]
        assert got == c["emissions"]


def test_chunk_walk_edges():
    from offline_tarteel_amd.streaming import split_chunks

    a = np.arange(48000 + 9000, dtype=np.float32)
    ch = split_chunks(a, 3.0, 0.0)
    assert [len(c) for c in ch] == [48000, 16000]            # 9000-sample tail zero-padded to 1 s
    assert ch[1][8999] == a[-1] and ch[1][9000] == 0.0
    assert [len(c) for c in split_chunks(a[:48000 + 7999], 3.0, 0.0)] == [48000]   # < 0.5 s tail dropped
    assert len(split_chunks(a[:7000], 3.0, 0.0)) == 0
    assert [len(c) for c in split_chunks(a[:48000], 2.0, 0.5)] == [32000, 24000]               # step 1.5 s


def test_wav16_roundtrip(tmp_path):
    from offline_tarteel_amd.audio import load_audio
    from offline_tarteel_amd.streaming import _write_wav16

    x = np.linspace(-1.0, 1.0, 1601, dtype=np.float32)
    p = tmp_path / "c.wav"
    _write_wav16(str(p), x)
    y = load_audio(str(p))
    assert len(y) == len(x) and np.max(np.abs(y - x)) <= 2.0 / 32768.0   # scale 32767 out, 32768 in, half an LSB of rounding


@pytest.fixture(scope="module")
def fulltx(golden_dir):
    with gzip.open(golden_dir / "fulltx_cases.json.gz", "rt", encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def mv_oracle(oracle):
    from oracle.tracker_ref import MatchVerseOracle

    return MatchVerseOracle(oracle)


def test_oracle_match_verse_hint_matches_reference(fulltx, mv_oracle):
    """match_verse(text, max_span=8, hint) without the trigram restriction: winner, fp64 score,
    raw score, bonus and the matched text's word count equal the reference's."""
    for c in fulltx["match"]:
        r = mv_oracle.match_verse(c["text"], max_span=8, hint=tuple(c["hint"]) if c["hint"] else None)
        w = c["result"]
        assert (r is None) == (w is None), c["text"]
        if r:
            for k in ("surah", "ayah", "ayah_end", "score", "raw_score", "bonus", "n_words"):
                assert r[k] == w[k], (c["text"], k, r[k], w[k])


def test_host_run_on_full_transcript_matches_reference(fulltx, mv_oracle):
    from offline_tarteel_amd.streaming import StreamingPipeline

    pipe = StreamingPipeline(matcher=lambda *a: [], match_verse_fn=lambda text, max_span, hint:
                             mv_oracle.match_verse(text, max_span=max_span, hint=hint))
    for c in fulltx["full"]:
        assert pipe.run_on_full_transcript("x.wav", lambda p, t=c["text"]: t) == c["emissions"], c["text"]
