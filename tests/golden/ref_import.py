"""Import the *unmodified* reference post-logits path in THIS container.

Generation-time only.  Nothing under tests/ that runs with ``-m gpu`` (and
nothing in bench.py / smoke()) imports this module: /root/reference does not
exist on the GPU box.  It is used by gen_golden.py to emit the fixtures that
pin oracle/ (SURVEY.md section 8c).

Three third-party packages the reference imports are absent here and are
replaced before import:

* ``Levenshtein.ratio`` (python-Levenshtein 0.27.3 -> rapidfuzz 3.14.3,
  uv.lock:1545,3674): normalised Indel similarity
  ``2*LCS(a,b) / (len(a)+len(b))`` (1.0 for two empty strings).  Restated here
  with arbitrary-precision-integer bit-parallel LCS.  Its own known answers
  are committed in tests/golden/indel_known_answers.json and checked by
  tests/test_oracle_text.py.
* ``librosa`` / ``soundfile``: only touched by shared/audio.py::load_audio,
  which the post-logits path never calls.  Empty stand-in modules.
* NeMo's tokenizer (nemo-toolkit 2.7.0, uv.lock:2313):
  ``text_to_ids`` / ``ids_to_text`` of the non-legacy SentencePieceTokenizer
  are ``encode_as_ids`` / ``decode_ids`` of the SentencePiece model shipped at
  web/frontend/public/tokenizer.model (sentencepiece is installed here).
"""

from __future__ import annotations

import importlib.util
import sys
import types
from pathlib import Path

REF = Path("/root/reference")


def indel_ratio(a: str, b: str) -> float:
    la, lb = len(a), len(b)
    if la + lb == 0:
        return 1.0
    if la == 0 or lb == 0:
        return 0.0
    # Hyyro/Crochemore bit-vector LCS with Python big ints.
    pm: dict[str, int] = {}
    for i, ch in enumerate(a):
        pm[ch] = pm.get(ch, 0) | (1 << i)
    full = (1 << la) - 1
    v = full
    for ch in b:
        m = pm.get(ch, 0)
        u = v & m
        v = ((v + u) | (v & ~m)) & full
    lcs = la - bin(v).count("1")
    dist = la + lb - 2 * lcs
    # rapidfuzz: norm_sim = 1 - dist/maximum  (maximum = la+lb)
    return 1.0 - dist / (la + lb)


class _Tok:
    def __init__(self):
        import sentencepiece as spm

        self.sp = spm.SentencePieceProcessor(
            model_file=str(REF / "web/frontend/public/tokenizer.model")
        )

    def text_to_ids(self, text):
        return self.sp.encode_as_ids(text)

    def ids_to_text(self, ids):
        return self.sp.decode_ids([int(i) for i in ids])


class _Model:
    def __init__(self):
        self.tokenizer = _Tok()


def load_reference():
    """Returns (c2c_direct_module, tokenizer) with _model/_db populated."""
    lev = types.ModuleType("Levenshtein")
    lev.ratio = indel_ratio
    sys.modules["Levenshtein"] = lev
    for name in ("librosa", "soundfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))
    spec = importlib.util.spec_from_file_location(
        "_c2c_direct_ref", str(REF / "experiments/c2c-direct/run.py")
    )
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod._model = _Model()
    mod._db = mod.QuranDB()
    return mod
