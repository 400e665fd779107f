"""Python face of the CPU oracle (post-logits stages).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  Pinned against the reference through
tests/golden/* (see tests/test_oracle_*.py); heavy integer work lives in qv_oracle.c.

Stages (reference file:line in each docstring):
    greedy_ids / ids_to_text / greedy_decode     c2c-direct/run.py:187-204
    normalize_arabic                             shared/normalizer.py:45-94
    match_verse / search / pass3 / build_candidates / ctc_rerank / predict
"""

from __future__ import annotations

import ctypes as C
import math
import re
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
DEFAULT_TABLES = ROOT / "offline-tarteel_amd" / "data" / "qverse_tables.bin"

VOCAB = 1025
BLANK = 1024
OTHER = 63


# ----------------------------------------------------------------- normaliser ----
# Independent restatement of shared/normalizer.py:45-94 (regex pipeline, default flags);
# the product has its own single-pass implementation, the tests compare both to the goldens.
def normalize_arabic(text: str) -> str:
    t = str(text).replace("﻿", "").replace("‏", "").replace("‎", "")
    t = re.sub("[ً-ٟ]", "", t)
    t = t.replace("آ", "ا").replace("ٱ", "ا")
    t = re.sub("[ٲٳ]", "ا", t)
    t = re.sub("اٰ", "ا", t)
    t = t.replace("ٰ", "ا")
    t = re.sub("[یے]", "ي", t)
    t = t.replace("ک", "ك")
    t = re.sub("[ۖ-ۭ]", "", t)
    t = re.sub("[﴾﴿]", "", t)
    t = re.sub("[٠-٩۰-۹]", "", t)
    t = t.replace("ـ", "")
    t = re.sub("[.,;:!?…،؛؟]", "", t)
    return re.sub(r"\s+", " ", t).strip()


# ----------------------------------------------------------------- tables --------
def read_blob(path) -> dict[str, np.ndarray]:
    raw = np.fromfile(str(path), dtype=np.uint8)
    assert raw[:8].tobytes() == b"QVTB0001"
    n = int(raw[8:12].view(np.uint32)[0])
    dtypes = {
        "meta": np.int32, "alphabet": np.uint32, "surah": np.uint8, "ayah": np.uint16,
        "surah_start": np.int32, "surah_len": np.int32, "tok": np.uint16, "vtri": np.uint16,
        "tri_keys": np.uint32, "tri_idf": np.float64, "clean_nw": np.uint16, "alt_nw": np.uint16,
        "nobsm_nw": np.uint16,
    }
    out = {}
    for i in range(n):
        e = raw[16 + 40 * i: 16 + 40 * (i + 1)]
        name = e[:24].tobytes().rstrip(b"\0").decode()
        off, nb = (int(x) for x in e[24:40].view(np.uint64))
        dt = dtypes.get(name, np.uint32 if name.endswith("_off") else np.uint8)
        out[name] = raw[off: off + nb].view(dt)
    return out


class _Match(C.Structure):
    _fields_ = [
        ("start", C.c_int32), ("span", C.c_int32), ("score", C.c_double), ("raw_score", C.c_double),
        ("n_runners", C.c_int32), ("runner_idx", C.c_int32 * 128), ("runner_score", C.c_double * 128),
    ]


class _Knobs(C.Structure):
    _fields_ = [("top_text", C.c_int32), ("top_span_refs", C.c_int32), ("max_span", C.c_int32)]


class _Result(C.Structure):
    _fields_ = [
        ("surah", C.c_int32), ("ayah", C.c_int32), ("ayah_end", C.c_int32), ("source", C.c_int32),
        ("score", C.c_double), ("ctc_norm_loss", C.c_float), ("n_candidates", C.c_int32),
        ("use_ctc", C.c_int32), ("base_score", C.c_double),
    ]


_U8P = C.POINTER(C.c_uint8)
_I32P = C.POINTER(C.c_int32)
_F64P = C.POINTER(C.c_double)
_F32P = C.POINTER(C.c_float)
_U16P = C.POINTER(C.c_uint16)


class Oracle:
    """CPU oracle over the static tables."""

    def __init__(self, tables=DEFAULT_TABLES, top_text=100, top_span_refs=80, max_span=6,
                 threshold=0.80, text_weight=0.0, span_penalty=0.5):
        import importlib.util

        spec = importlib.util.spec_from_file_location("_qvo_build", str(HERE / "build.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        self.lib = C.CDLL(str(b.build()))
        L = self.lib
        L.qvo_open.restype = C.c_void_p
        L.qvo_open.argtypes = [C.c_char_p]
        L.qvo_lcs.restype = C.c_int
        L.qvo_lcs.argtypes = [_U8P, C.c_int, _U8P, C.c_int]
        for f in (L.qvo_ratio, L.qvo_partial_ratio, L.qvo_fragment_score):
            f.restype = C.c_double
            f.argtypes = [_U8P, C.c_int, _U8P, C.c_int]
        L.qvo_trigram_candidates.restype = C.c_int
        L.qvo_trigram_candidates.argtypes = [C.c_void_p, _U8P, C.c_int, C.c_int, _I32P]
        L.qvo_pyset_order.restype = C.c_int
        L.qvo_pyset_order.argtypes = [_I32P, C.c_int, _I32P]
        L.qvo_match_verse.restype = None
        L.qvo_match_verse.argtypes = [C.c_void_p, _U8P, C.c_int, C.c_int, C.c_int, C.POINTER(_Match)]
        for f in (L.qvo_search, L.qvo_pass3):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, _U8P, C.c_int, C.c_int, _I32P, _F64P]
        L.qvo_build_candidates.restype = C.c_int
        L.qvo_build_candidates.argtypes = [C.c_void_p, _U8P, C.c_int, _U8P, C.c_int, C.POINTER(_Knobs), _I32P, _I32P,
                                           _F64P, C.POINTER(_Match)]
        L.qvo_ctc_loss.restype = C.c_float
        L.qvo_ctc_loss.argtypes = [_F32P, C.c_int, C.c_int, _U16P, C.c_int, C.c_int]
        L.qvo_ctc_rerank.restype = C.c_int
        L.qvo_ctc_rerank.argtypes = [C.c_void_p, _F32P, C.c_int, C.c_int, C.c_int, _I32P, _I32P, _F64P,
                                     C.c_double, C.c_double, _F32P, _I32P, _F64P]
        L.qvo_predict_from_transcript.restype = None
        L.qvo_predict_from_transcript.argtypes = [C.c_void_p, _U8P, C.c_int, _F32P, C.c_int, C.c_int,
                                                  C.POINTER(_Knobs), C.c_double, C.c_double, C.c_double,
                                                  C.POINTER(_Result)]
        L.qvo_num_trigrams.argtypes = [C.c_void_p]
        L.qvo_num_postings.argtypes = [C.c_void_p]
        self.db = L.qvo_open(str(tables).encode())
        if not self.db:
            raise FileNotFoundError(f"cannot open tables {tables}")
        self.t = read_blob(tables)
        self.knobs = _Knobs(top_text, top_span_refs, max_span)
        self.threshold, self.text_weight, self.span_penalty = threshold, text_weight, span_penalty
        self.alphabet = [chr(int(c)) for c in self.t["alphabet"]]
        self.code_of = {ch: i for i, ch in enumerate(self.alphabet)}
        po, pu = self.t["piece_u8_off"], self.t["piece_u8"]
        self.piece_surface = [pu[po[i]: po[i + 1]].tobytes().decode("utf-8") for i in range(VOCAB)]
        self.surah = self.t["surah"]
        self.ayah = self.t["ayah"]

    # ------------------------------------------------------------- strings -------
    def encode(self, s: str) -> np.ndarray:
        return np.array([self.code_of.get(ch, OTHER) for ch in s], dtype=np.uint8)

    @staticmethod
    def _p(a: np.ndarray, typ):
        return a.ctypes.data_as(typ)

    def verse_index(self, surah: int, ayah: int) -> int:
        return int(self.t["surah_start"][surah - 1]) + ayah - 1

    def key_of(self, start: int, span: int):
        s, a = int(self.surah[start]), int(self.ayah[start])
        return (s, a, a + span - 1)

    def verse_text(self, idx: int, which: str = "clean") -> str:
        off, codes = self.t[which + "_off"], self.t[which]
        return "".join(self.alphabet[c] for c in codes[off[idx]: off[idx + 1]])

    def token_ids(self, start: int, span: int) -> np.ndarray:
        k = start * 6 + (span - 1)
        return self.t["tok"][self.t["tok_off"][k]: self.t["tok_off"][k + 1]]

    # ------------------------------------------------------------- decode --------
    @staticmethod
    def greedy_ids(log_probs: np.ndarray) -> list[int]:
        """argmax per frame, drop repeats and blank (c2c-direct/run.py:193-200)."""
        ids = np.asarray(log_probs).argmax(-1)
        out, prev = [], -1
        for i in ids.tolist():
            if i != prev and i != BLANK:
                out.append(i)
            prev = i
        return out

    def ids_to_text(self, ids) -> str:
        """SentencePiece decode_ids: concatenate piece surfaces ('▁' -> ' ', unk -> ' ⁇ '),
        drop the dummy-prefix space (pinned by tests/golden/tokenizer_cases.json)."""
        out = ""
        for i in ids:
            surf = self.piece_surface[int(i)]
            # leading-whitespace pieces at the start of the text lose their one leading
            # marker (SentencePiece decode with remove_extra_whitespaces); <unk> keeps " ⁇ "
            if not out and int(i) != 0 and surf.startswith(" "):
                surf = surf[1:]
            out += surf
        return out

    def greedy_decode(self, log_probs: np.ndarray) -> str:
        ids = self.greedy_ids(log_probs)
        if not ids:
            return ""
        return normalize_arabic(self.ids_to_text(ids).strip())

    # ------------------------------------------------------------- retrieval -----
    def lcs(self, a: str, b: str) -> int:
        ea, eb = self.encode(a), self.encode(b)
        return self.lib.qvo_lcs(self._p(ea, _U8P), len(ea), self._p(eb, _U8P), len(eb))

    def lcs_raw(self, a: str, b: str) -> int:
        """LCS over arbitrary strings (own private alphabet) -- for the known-answer tests."""
        chars = {ch: i for i, ch in enumerate(sorted(set(a) | set(b)))}
        assert len(chars) < OTHER
        ea = np.array([chars[c] for c in a], dtype=np.uint8)
        eb = np.array([chars[c] for c in b], dtype=np.uint8)
        return self.lib.qvo_lcs(self._p(ea, _U8P), len(ea), self._p(eb, _U8P), len(eb))

    def ratio_raw(self, a: str, b: str) -> float:
        la, lb = len(a), len(b)
        if la + lb == 0:
            return 1.0
        return 1.0 - (la + lb - 2 * self.lcs_raw(a, b)) / (la + lb)

    def trigram_candidates(self, text: str, top_k=50) -> list[int]:
        q = self.encode(text)
        out = np.zeros(max(top_k, 1), dtype=np.int32)
        n = self.lib.qvo_trigram_candidates(self.db, self._p(q, _U8P), len(q), top_k, self._p(out, _I32P))
        return out[:n].tolist()

    def pyset_order(self, vals) -> list[int]:
        v = np.asarray(vals, dtype=np.int32)
        out = np.zeros(len(v) + 1, dtype=np.int32)
        n = self.lib.qvo_pyset_order(self._p(v, _I32P), len(v), self._p(out, _I32P))
        return out[:n].tolist()

    def match_verse(self, text: str, top_k=None):
        q = self.encode(normalize_arabic(text))
        m = _Match()
        self.lib.qvo_match_verse(self.db, self._p(q, _U8P), len(q), self.knobs.max_span,
                                 self.knobs.top_text if top_k is None else top_k, C.byref(m))
        if m.start < 0:
            return None
        s, a, e = self.key_of(m.start, m.span)
        return {
            "surah": s, "ayah": a, "ayah_end": e if m.span > 1 else None, "score": m.score,
            "raw_score": m.raw_score, "start": m.start, "span": m.span,
            "runners_up": [
                (int(self.surah[m.runner_idx[i]]), int(self.ayah[m.runner_idx[i]]), m.runner_score[i])
                for i in range(m.n_runners)
            ],
        }

    def _topk(self, fn, text, top_k):
        q = self.encode(text)
        idx = np.zeros(top_k, dtype=np.int32)
        sc = np.zeros(top_k, dtype=np.float64)
        n = fn(self.db, self._p(q, _U8P), len(q), top_k, self._p(idx, _I32P), self._p(sc, _F64P))
        return [(int(self.surah[i]), int(self.ayah[i]), float(s)) for i, s in zip(idx[:n], sc[:n])]

    def search(self, text: str, top_k=100):
        return self._topk(self.lib.qvo_search, normalize_arabic(text), top_k)

    def pass3(self, text: str, top_k=100):
        return self._topk(self.lib.qvo_pass3, text, top_k)

    def build_candidates(self, transcript: str):
        """transcript is used raw by pass 3 and normalised by match_verse/search
        (c2c-direct/run.py:258-297 vs quran_db.py:93,268)."""
        q = self.encode(transcript)
        qn = self.encode(normalize_arabic(transcript))
        cs = np.zeros(8192, dtype=np.int32)
        cp = np.zeros(8192, dtype=np.int32)
        sc = np.zeros(8192, dtype=np.float64)
        m = _Match()
        n = self.lib.qvo_build_candidates(self.db, self._p(q, _U8P), len(q), self._p(qn, _U8P), len(qn),
                                          C.byref(self.knobs),
                                          self._p(cs, _I32P), self._p(cp, _I32P), self._p(sc, _F64P),
                                          C.byref(m))
        return cs[:n].copy(), cp[:n].copy(), sc[:n].copy(), m

    # ------------------------------------------------------------- CTC -----------
    def ctc_loss_c(self, lp: np.ndarray, ids) -> float:
        lp = np.ascontiguousarray(lp, dtype=np.float32)
        ids = np.ascontiguousarray(ids, dtype=np.uint16)
        return float(self.lib.qvo_ctc_loss(self._p(lp, _F32P), lp.shape[0], lp.shape[1],
                                           self._p(ids, _U16P), len(ids), BLANK))

    @staticmethod
    def ctc_loss_torch(lp: np.ndarray, id_lists, batch=16) -> np.ndarray:
        """The reference's own call (c2c-direct/run.py:341-362): batches of 16, expanded
        log-probs, F.ctc_loss(reduction='none', zero_infinity=True) on CPU float32."""
        import torch
        import torch.nn.functional as F

        t, vocab = lp.shape
        log_probs = torch.from_numpy(np.ascontiguousarray(lp, dtype=np.float32)).unsqueeze(1)
        out = []
        for s in range(0, len(id_lists), batch):
            chunk = id_lists[s: s + batch]
            n = len(chunk)
            targets = torch.tensor([int(x) for seq in chunk for x in seq], dtype=torch.long)
            tl = torch.tensor([len(seq) for seq in chunk], dtype=torch.long)
            il = torch.full((n,), t, dtype=torch.long)
            losses = F.ctc_loss(log_probs.expand(t, n, vocab).contiguous(), targets, il, tl,
                                blank=BLANK, reduction="none", zero_infinity=True)
            out.extend(losses.tolist())
        return np.asarray(out, dtype=np.float32)

    @staticmethod
    def ctc_score_f64(lp: np.ndarray, ids) -> float:
        """float64 restatement of the browser twin of the rerank, web/frontend/src/lib/
        ctc-rescore.ts:35-102 (scoreCtcSequence): alpha recursion over the 2L+1 blank-interleaved
        states with logAddExp = hi + log1p(exp(lo - hi)) (:22-28), infeasible (L == 0 or
        2L+1 > T, :30-33,43-49) -> 1e9, else -log p / L.  A cross-check of the float32 HIP
        kernel in double precision; vectorised over the states."""
        lp = np.asarray(lp, dtype=np.float64)
        T, L = lp.shape[0], len(ids)
        if L == 0 or 2 * L + 1 > T:
            return 1e9
        S = 2 * L + 1
        states = np.full(S, BLANK, dtype=np.int64)
        states[1::2] = np.asarray(ids, dtype=np.int64)
        skip = np.zeros(S, dtype=bool)
        skip[2:] = (states[2:] != BLANK) & (states[2:] != states[:-2])

        def lae(a, b):
            hi, lo = np.maximum(a, b), np.minimum(a, b)
            with np.errstate(invalid="ignore"):
                r = hi + np.log1p(np.exp(lo - hi))
            return np.where(np.isneginf(lo), hi, r)

        prev = np.full(S, -np.inf)
        prev[0] = lp[0, BLANK]
        prev[1] = lp[0, states[1]]
        for t in range(1, T):
            tot = prev.copy()
            tot[1:] = lae(tot[1:], prev[:-1])
            cand = np.full(S, -np.inf)
            cand[2:] = prev[:-2]
            tot = np.where(skip, lae(tot, cand), tot)
            prev = np.where(np.isneginf(tot), -np.inf, tot + lp[t, states])
        fin = float(lae(prev[S - 1], prev[S - 2]))
        if not np.isfinite(fin):
            return 1e9
        return -fin / L

    def ctc_rerank(self, lp: np.ndarray, cs, cp, sc):
        lp = np.ascontiguousarray(lp, dtype=np.float32)
        n = len(cs)
        loss = np.zeros(n, dtype=np.float32)
        cl = np.zeros(n, dtype=np.int32)
        fs = np.zeros(n, dtype=np.float64)
        cs = np.ascontiguousarray(cs, dtype=np.int32)
        cp = np.ascontiguousarray(cp, dtype=np.int32)
        sc = np.ascontiguousarray(sc, dtype=np.float64)
        win = self.lib.qvo_ctc_rerank(self.db, self._p(lp, _F32P), lp.shape[0], lp.shape[1], n,
                                      self._p(cs, _I32P), self._p(cp, _I32P), self._p(sc, _F64P),
                                      self.text_weight, self.span_penalty,
                                      self._p(loss, _F32P), self._p(cl, _I32P), self._p(fs, _F64P))
        return win, loss, cl, fs

    # ------------------------------------------------------------- predict -------
    def predict_logprobs(self, lp: np.ndarray) -> dict:
        """experiments/c2c-direct-mixed/run.py:66-133 from the [T,1025] log-probs on."""
        lp = np.ascontiguousarray(lp, dtype=np.float32)
        ids = self.greedy_ids(lp)
        transcript = normalize_arabic(self.ids_to_text(ids).strip()) if ids else ""
        if not transcript.strip():
            return {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0, "transcript": "",
                    "source": None, "greedy_ids": ids}
        q = self.encode(transcript)
        r = _Result()
        self.lib.qvo_predict_from_transcript(self.db, self._p(q, _U8P), len(q), self._p(lp, _F32P),
                                             lp.shape[0], lp.shape[1], C.byref(self.knobs),
                                             self.threshold, self.text_weight, self.span_penalty,
                                             C.byref(r))
        if r.source == 0:
            return {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0, "transcript": transcript,
                    "source": None, "greedy_ids": ids}
        return {
            "surah": r.surah, "ayah": r.ayah, "ayah_end": r.ayah_end,
            "score": round(r.score, 4), "score_raw": r.score,
            "transcript": transcript, "source": "ctc" if r.source == 2 else "text",
            "greedy_ids": ids, "n_candidates": r.n_candidates, "use_ctc": bool(r.use_ctc),
            "ctc_norm_loss": float(r.ctc_norm_loss), "base_score": r.base_score,
        }


def score_sequence(expected, predicted):
    """benchmark/runner.py:104-143 (ordered subsequence recall / precision / exact)."""
    if not expected:
        return {"recall": 1.0, "precision": 1.0, "sequence_accuracy": 1.0}
    if not predicted:
        return {"recall": 0.0, "precision": 0.0, "sequence_accuracy": 0.0}
    pt = [(p["surah"], p["ayah"]) for p in predicted]
    et = [(e["surah"], e["ayah"]) for e in expected]
    hit, j0, used = 0, 0, set()
    for e in et:
        for j in range(j0, len(pt)):
            if pt[j] == e:
                hit += 1
                used.add(j)
                j0 = j + 1
                break
    return {"recall": hit / len(et), "precision": len(used) / len(pt),
            "sequence_accuracy": 1.0 if pt == et else 0.0}
