"""CPU restatement of the verse tracker's matching step (TEST INFRASTRUCTURE ONLY).

Follows shared/verse_tracker.py of the reference:

    _score_verse       :41-65    prefix / full Indel ratios, coverage blend, continuation bonus
    _find_best_match   :67-101   scan of all verses (clean text, then the bismillah-stripped
                                 variant when the verse has one), strict ">" so the first
                                 maximum in verse order wins, minimum-score and word-count gates
    get_next_verse     shared/quran_db.py:81-90

Pinned by tests/golden/tracker_cases.json.gz, which tests/golden/gen_tracker_golden.py
produced by running the unmodified reference classes in the build container.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(offline-tarteel_amd/verse_tracker.py) calls the HIP library and has no CPU path.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from .oracle import Oracle, normalize_arabic  # noqa: F401

_U8P = C.POINTER(C.c_uint8)

CONTINUATION_BONUS = 0.15      # verse_tracker.py:14
MIN_EMIT_SCORE = 0.3           # :16
STREAMING_MIN_EMIT_SCORE = 0.4  # :18
MIN_WORDS_FOR_MATCH = 2        # :19


class TrackerOracle:
    def __init__(self, oracle: Oracle | None = None):
        self.o = oracle or Oracle()
        t = self.o.t
        self.n = len(t["surah"])
        self.texts = []          # per verse: [(codes, word_ends, variant)] in the reference's order
        for v in range(self.n):
            clean = np.ascontiguousarray(t["clean"][t["clean_off"][v]: t["clean_off"][v + 1]])
            entry = [(clean, self._word_ends(clean), 0)]
            nb = np.ascontiguousarray(t["nobsm"][t["nobsm_off"][v]: t["nobsm_off"][v + 1]])
            if len(nb):
                entry.append((nb, self._word_ends(nb), 2))
            self.texts.append(entry)

    @staticmethod
    def _word_ends(codes: np.ndarray) -> list[int]:
        """Character count of the first k words joined by single spaces, k = 1..n_words."""
        sp = np.flatnonzero(codes == 0).tolist()
        return sp + [len(codes)]

    def next_verse(self, last):
        """quran_db.py:81-90: the verse after (surah, ayah) in mushaf order, None after 114:6 or
        when (surah, ayah) does not exist."""
        if not last:
            return None
        s, a = last
        t = self.o.t
        if not (1 <= s <= 114) or not (1 <= a <= int(t["surah_len"][s - 1])):
            return None
        idx = int(t["surah_start"][s - 1]) + a - 1
        return idx + 1 if idx + 1 < self.n else None

    def _lcs(self, a: np.ndarray, b: np.ndarray, nb: int) -> int:
        return self.o.lib.qvo_lcs(a.ctypes.data_as(_U8P), len(a), b.ctypes.data_as(_U8P), nb)

    @staticmethod
    def _ratio(lcs: int, la: int, lb: int) -> float:
        if la + lb == 0:
            return 1.0
        return 1.0 - (la + lb - 2 * lcs) / (la + lb)

    def score_verse(self, q: np.ndarray, n_text: int, codes, ends, bonus: bool) -> float:
        n_verse = len(ends)
        p = min(n_text, n_verse)
        plen = ends[p - 1] if p > 0 else 0
        prefix_score = self._ratio(self._lcs(q, codes, plen), len(q), plen)
        full_score = self._ratio(self._lcs(q, codes, len(codes)), len(q), len(codes))
        coverage = n_text / max(n_verse, 1)
        if coverage > 0.8:
            raw = 0.3 * prefix_score + 0.7 * full_score
        else:
            raw = 0.7 * prefix_score + 0.3 * full_score
        if bonus:
            raw += CONTINUATION_BONUS
        return raw

    def best_raw(self, text: str, last=None):
        """(verse index, variant, n_words of the matched text, score) of the running maximum
        of the scan, before the minimum-score gate; None if no verse scores above 0."""
        q = np.ascontiguousarray(self.o.encode(text))
        n_text = len(text.split())
        nxt = self.next_verse(last)
        best, best_score = None, 0.0
        for v in range(self.n):
            score, var, nw = None, 0, 0
            for codes, ends, variant in self.texts[v]:
                s = self.score_verse(q, n_text, codes, ends, v == nxt)
                if score is None or s > score:
                    score, var, nw = s, variant, len(ends)
            if score > best_score:
                best_score = score
                best = (v, var, nw, score)
        return best

    def find_best_match(self, text: str, last=None, streaming: bool = False):
        if not text.strip():
            return None
        if streaming and len(text.split()) < MIN_WORDS_FOR_MATCH:
            return None
        b = self.best_raw(text, last)
        if b is None:
            return None
        v, var, nw, score = b
        if score < (STREAMING_MIN_EMIT_SCORE if streaming else MIN_EMIT_SCORE):
            return None
        return {"surah": int(self.o.surah[v]), "ayah": int(self.o.ayah[v]), "n_words": nw, "score": score,
                "verse": v, "variant": var}
