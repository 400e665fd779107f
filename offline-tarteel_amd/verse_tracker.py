"""Verse tracker for streaming transcripts -- host-side mirror of the reference's
``shared/verse_tracker.py`` (class VerseTracker, :22-244) over the HIP matching step.

Same constructor arguments, same three calls (``process_text`` = the whole transcript so far,
``process_delta`` = one more chunk, ``finalize``), same emissions ``{"surah", "ayah", "score"}``.
What the reference does in ``_find_best_match`` -- a Python loop over all 6,236 verses with two
Indel ratios each, ~0.2 s per call -- is one launch of ``qv_tracker_match`` here
(include/qverse.h), and it is the only thing this module computes off the host: there is no CPU
matching path, a tracker without a working library raises.

The state machine is written as a generator that *yields* each matching request
``(text, last_emitted)`` and is resumed with the raw match.  One tracker drives it with
single-text launches; ``drive_many`` advances any number of trackers in lock step and answers all
their pending requests with ONE batched launch per round, which is how the streaming pipeline
follows many recordings at once.
"""

from __future__ import annotations

from .normalizer import normalize_arabic

CONTINUATION_BONUS = 0.15        # applied on the device (verse_tracker.py:14)
SCORE_DROP_THRESHOLD = 0.15      # :15
MIN_EMIT_SCORE = 0.3             # :16
OVERFLOW_RATIO = 1.15            # :17
STREAMING_MIN_EMIT_SCORE = 0.4   # :18
MIN_WORDS_FOR_MATCH = 2          # :19


def _default_matcher():
    from .plugin import _ensure_engine

    return _ensure_engine().track_match


class VerseTracker:
    """``db`` is accepted for signature compatibility with the reference (a QuranDB there); here it
    may be an ``Engine`` (its ``track_match`` is used) or None (the plugin's process-wide engine).
    ``matcher(texts, last_refs) -> [match | None]`` overrides both (the parity tests inject the
    CPU oracle there to check this host logic against the golden emissions)."""

    def __init__(self, db=None, last_emission: tuple[int, int] | None = None, streaming_mode: bool = False,
                 matcher=None):
        if matcher is None:
            matcher = db.track_match if db is not None and hasattr(db, "track_match") else _default_matcher()
        self._matcher = matcher
        self._streaming_mode = streaming_mode
        self._min_emit_score = STREAMING_MIN_EMIT_SCORE if streaming_mode else MIN_EMIT_SCORE
        self._accumulated = ""
        self._current_match: dict | None = None
        self._peak_score = 0.0
        self._emissions: list[dict] = []
        self._last_emitted = last_emission

    # ------------------------------------------------------------ generator core ----
    def _find(self, text: str):
        """verse_tracker.py:67-101.  Yields one request unless a host-side gate answers first."""
        if not text.strip():
            return None
        if self._streaming_mode and len(text.split()) < MIN_WORDS_FOR_MATCH:
            return None
        raw = yield (text, self._last_emitted)
        if raw is None or raw["score"] < self._min_emit_score:
            return None
        return {"surah": raw["surah"], "ayah": raw["ayah"], "n_words": raw["n_words"], "score": raw["score"]}

    def _emit(self, match: dict) -> dict | None:
        """:103-127 -- the accumulator always loses the matched verse's word count, duplicates
        of the last emission are swallowed."""
        acc = self._accumulated.split()
        self._accumulated = " ".join(acc[min(match["n_words"], len(acc)):])
        self._current_match, self._peak_score = None, 0.0
        ref = (match["surah"], match["ayah"])
        if ref == self._last_emitted:
            return None
        emission = {"surah": match["surah"], "ayah": match["ayah"], "score": match["score"]}
        self._emissions.append(emission)
        self._last_emitted = ref
        return emission

    def _split(self, match: dict):
        """:129-150 -- more words than the matched verse holds: emit it and look at the rest."""
        out = []
        if match["n_words"] > 0 and len(self._accumulated.split()) > match["n_words"] * OVERFLOW_RATIO:
            e = self._emit(match)
            if e:
                out.append(e)
            if self._accumulated.strip():
                nxt = yield from self._find(self._accumulated)
                if nxt:
                    more = yield from self._split(nxt)
                    if more:
                        out.extend(more)
                    else:
                        self._current_match, self._peak_score = nxt, nxt["score"]
        return out

    def _evaluate(self):
        """:152-205."""
        out = []
        match = yield from self._find(self._accumulated)
        if not match:
            return out
        cur = self._current_match
        if cur and (cur["surah"], cur["ayah"]) == (match["surah"], match["ayah"]):
            if match["score"] > self._peak_score:
                self._peak_score = match["score"]
            elif self._peak_score - match["score"] > SCORE_DROP_THRESHOLD:
                e = self._emit(cur)
                if e:
                    out.append(e)
                if self._accumulated.strip():
                    nxt = yield from self._find(self._accumulated)
                    if nxt:
                        self._current_match, self._peak_score = nxt, nxt["score"]
                    else:
                        self._current_match, self._peak_score = None, 0.0
            else:
                self._current_match = match
        else:
            if cur and cur["score"] >= self._min_emit_score:
                e = self._emit(cur)
                if e:
                    out.append(e)
            self._current_match, self._peak_score = match, match["score"]
        if not self._current_match:
            self._current_match, self._peak_score = match, match["score"]
        if self._current_match and not out:
            out.extend((yield from self._split(self._current_match)))
        return out

    def text_steps(self, text: str):
        """Generator form of process_text (:207-223)."""
        normalized = normalize_arabic(text)
        if not normalized.strip():
            return []
        self._accumulated = normalized
        return (yield from self._evaluate())

    def delta_steps(self, new_text: str):
        """Generator form of process_delta (:225-244)."""
        normalized = normalize_arabic(new_text)
        if not normalized.strip():
            return []
        self._accumulated = (self._accumulated + " " + normalized) if self._accumulated else normalized
        return (yield from self._evaluate())

    # ------------------------------------------------------------ reference API ------
    def _run(self, gen) -> list[dict]:
        return drive_many([gen], self._matcher)[0]

    def process_text(self, text: str) -> list[dict]:
        return self._run(self.text_steps(text))

    def process_delta(self, new_text: str) -> list[dict]:
        return self._run(self.delta_steps(new_text))

    def finalize(self) -> list[dict]:
        """:246-251"""
        cur = self._current_match
        if cur and cur["score"] >= self._min_emit_score:
            e = self._emit(cur)
            return [e] if e else []
        return []


def drive_many(gens, matcher) -> list:
    """Run generators produced by ``text_steps`` / ``delta_steps`` (of DIFFERENT trackers) to
    completion; each round gathers the pending requests of all still-running generators into one
    ``matcher(texts, last_refs)`` call.  Returns each generator's return value, in order."""
    results = [None] * len(gens)
    pending = {}
    for i, g in enumerate(gens):
        try:
            pending[i] = next(g)
        except StopIteration as stop:
            results[i] = stop.value
    while pending:
        order = list(pending)
        answers = matcher([pending[i][0] for i in order], [pending[i][1] for i in order])
        for i, ans in zip(order, answers):
            try:
                pending[i] = gens[i].send(ans)
            except StopIteration as stop:
                results[i] = stop.value
                del pending[i]
    return results
