"""Host-side Arabic text normalisation (same contract as the reference's
``shared/normalizer.py:45-94`` ``normalize_arabic`` with its default flags).

Written as a single code-point pass instead of a regex pipeline: every rule of the
default configuration is a per-character delete / map, except

* ``ALEF + SUPERSCRIPT ALEF -> ALEF`` (reference :66-67; the superscript alef on its
  own becomes a plain alef), handled with one character of look-behind *on the
  stream after diacritic removal*, exactly where the reference applies it, and
* whitespace collapsing + strip (reference :91-92).

``strip_hamza=True`` (off by default and never enabled on the c2c path) is supported
for API completeness and is applied in the reference's order.
"""

from __future__ import annotations

import re

_DROP_ALWAYS = {0xFEFF, 0x200F, 0x200E}
_ALEF = "ا"

_PUNCT = set(".,;:!?…،؛؟")


def _is_diacritic(o: int) -> bool:
    return 0x064B <= o <= 0x065F


def normalize_arabic(
    text: str,
    diacritics: bool = True,
    markers: bool = True,
    verse_numbers: bool = True,
    tatweel: bool = True,
    small_letters: bool = True,
    punctuation: bool = True,
    collapse_whitespace: bool = True,
    strip_hamza: bool = False,
) -> str:
    out: list[str] = []
    # stage A: everything up to and including the alef/khanjariya rules operates on
    # the diacritic-free stream, so do that as its own pass to keep look-behind exact.
    stage: list[str] = []
    for ch in str(text):
        o = ord(ch)
        if o in _DROP_ALWAYS:
            continue
        if diacritics:
            if _is_diacritic(o):
                continue
            if o in (0x0622, 0x0671, 0x0672, 0x0673):
                ch = _ALEF
        stage.append(ch)
    if diacritics:
        # pair rule then leftovers, both left-to-right and non-overlapping, as re.sub /
        # str.replace apply them in the reference (:66-67)
        joined = "".join(stage).replace("اٰ", _ALEF).replace("ٰ", _ALEF)
        stage = list(joined)
    for ch in stage:
        o = ord(ch)
        if diacritics:
            if o in (0x06CC, 0x06D2):
                ch, o = "ي", 0x064A
            elif o == 0x06A9:
                ch, o = "ك", 0x0643
        if (markers or small_letters) and 0x06D6 <= o <= 0x06ED:
            continue
        if verse_numbers and (o in (0xFD3E, 0xFD3F) or 0x0660 <= o <= 0x0669 or 0x06F0 <= o <= 0x06F9):
            continue
        if tatweel and o == 0x0640:
            continue
        if punctuation and ch in _PUNCT:
            continue
        out.append(ch)
    s = "".join(out)
    if strip_hamza:
        s = re.sub("[ءأإئ]", "", s)
        s = s.replace("ى", "ي")
        s = re.sub("وا?ة", "اة", s)
        s = re.sub("يي", "ي", s)
        s = s.replace("بصط", "بسط")
        s = s.replace("صيطر", "سيطر")
        s = re.sub("الل", "ال", s)
    if collapse_whitespace:
        s = " ".join(s.split())  # == re.sub(r"\s+", " ", s).strip() (both use str.isspace)
    return s
