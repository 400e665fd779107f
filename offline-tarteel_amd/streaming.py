"""Streaming pipeline -- host-side mirror of the reference's ``shared/streaming.py``
(class StreamingPipeline, :31-210) over the MI355X engine.

    run_on_text(snapshots)                 :36-56   accumulated-text snapshots -> emissions
    run_on_audio_chunked(path, fn, c, o)   :107-210 chunk walk, confidence gate, tentative /
                                                    confirmed emissions, VerseTracker in
                                                    streaming mode fed chunk by chunk
    run_on_audio_chunked_batch(paths)      (new)    the same for many recordings at once
    run_on_full_transcript(path, fn)       :58-105  whole transcript, verses peeled off front to back

The chunk walk never depends on a transcript, so with the engine as the ASR backend
(``transcribe_fn=None``) ALL chunks of ALL recordings go through the acoustic model as one packed
ragged batch, and the trackers are then advanced in lock step with one ``qv_tracker_match`` launch
per round (verse_tracker.drive_many).  A caller-supplied ``transcribe_fn(wav_path) -> str | dict``
is honoured exactly like in the reference (one 16-bit PCM temporary WAV per chunk).

``run_on_full_transcript`` (:58-105) peels verses off the front of a whole-file transcript with
``match_verse(remaining, max_span=8, hint=...)``; that call is ``qv_match_verse`` here (full scan of
all verses, continuation bonuses and suffix-prefix scores on the device).
"""

from __future__ import annotations

import os
import tempfile
import wave
from pathlib import Path

import numpy as np

from .audio import load_audio
from .normalizer import normalize_arabic
from .verse_tracker import STREAMING_MIN_EMIT_SCORE, VerseTracker, drive_many

SAMPLE_RATE = 16000
MIN_CHUNK_SAMPLES = 8000          # streaming.py:23 -- a shorter tail chunk ends the walk
MIN_CHUNK_LOG_PROB = -1.0         # :24
MIN_CHUNK_WORDS = 2               # :25
HIGH_CONFIDENCE_THRESHOLD = 0.7   # :26
MAX_HOLD_CHUNKS = 3               # :27


def split_chunks(audio: np.ndarray, chunk_seconds: float = 3.0, overlap_seconds: float = 0.0) -> list[np.ndarray]:
    """The chunk walk of run_on_audio_chunked (:130-146): fixed step, stop at the first chunk
    shorter than 0.5 s, zero-pad chunks shorter than 1 s to 1 s."""
    chunk = int(chunk_seconds * SAMPLE_RATE)
    step = max(chunk - int(overlap_seconds * SAMPLE_RATE), 1)
    out, pos = [], 0
    while pos < len(audio):
        c = audio[pos:min(pos + chunk, len(audio))]
        if len(c) < MIN_CHUNK_SAMPLES:
            break
        if len(c) < SAMPLE_RATE:
            c = np.pad(c, (0, SAMPLE_RATE - len(c)))
        out.append(np.ascontiguousarray(c, dtype=np.float32))
        pos += step
    return out


def _write_wav16(path: str, x: np.ndarray):
    """What soundfile.write(path, float_array, 16000) stores by default: 16-bit PCM."""
    pcm = np.clip(np.rint(np.asarray(x, np.float64) * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(SAMPLE_RATE)
        w.writeframes(pcm.tobytes())


class _ChunkState:
    """Per-recording confirmation state of run_on_audio_chunked (:135-137, :165-203)."""

    def __init__(self, tracker: VerseTracker):
        self.tracker = tracker
        self.confirmed: list[dict] = []
        self.tentative: dict | None = None
        self.tentative_age = 0

    def gate(self, raw):
        """-> chunk text to feed, or None when the chunk is skipped (:156-178)."""
        if isinstance(raw, dict):
            text = raw.get("text", "").strip()
            avg_logprob = raw.get("avg_logprob", 0.0)
            n_words = len(text.split()) if text else 0
            gated = avg_logprob < MIN_CHUNK_LOG_PROB or n_words < MIN_CHUNK_WORDS
        else:
            text = str(raw).strip() if raw else ""
            gated = False
        if gated or not text:
            if self.tentative is not None:
                self.tentative_age += 1
                if self.tentative_age >= MAX_HOLD_CHUNKS:
                    self.tentative, self.tentative_age = None, 0
            return None
        return text

    def absorb(self, emissions: list[dict]):
        """A valid chunk arrived (:183-197)."""
        if self.tentative is not None:
            self.confirmed.append(self.tentative)
            self.tentative, self.tentative_age = None, 0
        for e in emissions:
            if e["score"] >= HIGH_CONFIDENCE_THRESHOLD:
                self.confirmed.append(e)
            else:
                if self.tentative is not None:
                    self.confirmed.append(self.tentative)
                self.tentative, self.tentative_age = e, 0

    def finish(self) -> list[dict]:
        """:201-207"""
        if self.tentative is not None and self.tentative["score"] >= STREAMING_MIN_EMIT_SCORE:
            self.confirmed.append(self.tentative)
        self.confirmed.extend(self.tracker.finalize())
        return self.confirmed


class StreamingPipeline:
    """``db``: an ``Engine`` (or None for the plugin's process-wide engine).  ``matcher`` overrides
    the tracker's matching step (tests)."""

    def __init__(self, db=None, matcher=None, match_verse_fn=None):
        self._engine = db
        self._matcher = matcher
        self._match_verse_fn = match_verse_fn   # (text, max_span, hint) -> dict | None; tests only

    def _eng(self):
        if self._engine is None:
            from .plugin import _ensure_engine

            self._engine = _ensure_engine()
        return self._engine

    def _match_fn(self):
        return self._matcher if self._matcher is not None else self._eng().track_match

    def _tracker(self, **kw) -> VerseTracker:
        return VerseTracker(matcher=self._match_fn(), **kw)

    # ------------------------------------------------------------------ text ------
    def run_on_text(self, text_chunks: list[str]) -> list[dict]:
        tracker = self._tracker()
        out = []
        for text in text_chunks:
            out.extend(tracker.process_text(text))
        out.extend(tracker.finalize())
        return out

    def run_on_full_transcript(self, audio_path, transcribe_fn=None) -> list[dict]:
        """:58-105.  ``transcribe_fn(audio_path) -> str``; None = the engine."""
        if transcribe_fn is None:
            import torch

            eng = self._eng()
            audio = self._load(audio_path)
            transcript = eng.transcribe_batch(torch.from_numpy(audio[None, :]).cuda(eng.device), [len(audio)])[0]
        else:
            transcript = transcribe_fn(audio_path)
        match = self._match_verse_fn
        if match is None:
            eng = self._eng()
            match = lambda text, max_span, hint: eng.match_verse(text, max_span=max_span, hint=hint)  # noqa: E731
        remaining = normalize_arabic(transcript)
        if not remaining.strip():
            return []
        out, hint, min_score = [], None, 0.3
        for _ in range(20):                       # safety bound of the reference
            if not remaining.strip():
                break
            r = match(remaining, 8, hint)
            if not r or r.get("score", 0) < min_score:
                break
            min_score = 0.7                       # after the first match only confident ones continue
            end = r.get("ayah_end") or r["ayah"]
            out.extend({"surah": r["surah"], "ayah": a, "score": r["score"]} for a in range(r["ayah"], end + 1))
            words = remaining.split()
            remaining = " ".join(words[min(r["n_words"], len(words)):])
            hint = (r["surah"], end)
        return out

    # ------------------------------------------------------------------ audio -----
    def transcribe_chunks(self, chunk_lists: list[list[np.ndarray]]) -> list[list[str]]:
        """Every chunk of every recording through the engine, in packed ragged batches."""
        import torch

        eng = self._eng()
        flat = [c for chunks in chunk_lists for c in chunks]
        texts: list[str] = []
        cap = int(eng.max_batch)
        for i in range(0, len(flat), cap):
            part = flat[i:i + cap]
            n = max(len(c) for c in part)
            host = np.zeros((len(part), n), np.float32)
            for j, c in enumerate(part):
                host[j, :len(c)] = c
            dev = torch.from_numpy(host).cuda(eng.device)
            texts.extend(eng.transcribe_batch(dev, [len(c) for c in part]))
        out, k = [], 0
        for chunks in chunk_lists:
            out.append(texts[k:k + len(chunks)])
            k += len(chunks)
        return out

    @staticmethod
    def _call_fn(transcribe_fn, chunk: np.ndarray):
        tmp = tempfile.NamedTemporaryFile(suffix=".wav", delete=False)
        try:
            tmp.close()
            _write_wav16(tmp.name, chunk)
            return transcribe_fn(tmp.name)
        except Exception:
            return ""
        finally:
            os.unlink(tmp.name)

    @staticmethod
    def _load(audio) -> np.ndarray:
        if isinstance(audio, (str, Path)):
            return load_audio(str(audio))
        return np.asarray(audio, dtype=np.float32)

    def run_on_audio_chunked_batch(self, audio_paths, transcribe_fn=None, chunk_seconds: float = 3.0,
                                   overlap_seconds: float = 0.0) -> list[list[dict]]:
        """run_on_audio_chunked for several recordings (paths or float32 arrays) at once."""
        chunk_lists = [split_chunks(self._load(a), chunk_seconds, overlap_seconds) for a in audio_paths]
        if transcribe_fn is None:
            raws = self.transcribe_chunks(chunk_lists)
        states = [_ChunkState(self._tracker(streaming_mode=True)) for _ in chunk_lists]
        for k in range(max((len(c) for c in chunk_lists), default=0)):
            live, gens = [], []
            for i, chunks in enumerate(chunk_lists):
                if k >= len(chunks):
                    continue
                raw = raws[i][k] if transcribe_fn is None else self._call_fn(transcribe_fn, chunks[k])
                text = states[i].gate(raw)
                if text is not None:
                    live.append(i)
                    gens.append(states[i].tracker.delta_steps(text))
            for i, emissions in zip(live, drive_many(gens, self._match_fn())):
                states[i].absorb(emissions)
        return [s.finish() for s in states]

    def run_on_audio_chunked(self, audio_path, transcribe_fn=None, chunk_seconds: float = 3.0,
                             overlap_seconds: float = 0.0) -> list[dict]:
        return self.run_on_audio_chunked_batch([audio_path], transcribe_fn, chunk_seconds, overlap_seconds)[0]
