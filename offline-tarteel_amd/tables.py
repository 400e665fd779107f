"""Host-side view of qverse_tables.bin (see tools/build_tables.py for the layout)."""

from __future__ import annotations

from pathlib import Path

import numpy as np

_DTYPES = {
    "meta": np.int32, "alphabet": np.uint32, "surah": np.uint8, "ayah": np.uint16,
    "surah_start": np.int32, "surah_len": np.int32, "tok": np.uint16, "vtri": np.uint16,
    "tri_keys": np.uint32, "tri_idf": np.float64, "clean_nw": np.uint16, "alt_nw": np.uint16,
    "nobsm_nw": np.uint16,
}
OTHER = 63


class Tables:
    def __init__(self, path):
        raw = np.fromfile(str(Path(path)), dtype=np.uint8)
        if raw[:8].tobytes() != b"QVTB0001":
            raise ValueError(f"{path}: not a qverse tables file")
        n = int(raw[8:12].view(np.uint32)[0])
        self.s = {}
        for i in range(n):
            e = raw[16 + 40 * i: 16 + 40 * (i + 1)]
            name = e[:24].tobytes().rstrip(b"\0").decode()
            off, nb = (int(x) for x in e[24:40].view(np.uint64))
            dt = _DTYPES.get(name, np.uint32 if name.endswith("_off") else np.uint8)
            self.s[name] = raw[off: off + nb].view(dt)
        self.n_verses = int(self.s["meta"][0])
        self.alphabet = [chr(int(c)) for c in self.s["alphabet"]]
        self._code = {ch: i for i, ch in enumerate(self.alphabet)}
        top = max(int(c) for c in self.s["alphabet"])
        self._lut = np.full(top + 2, OTHER, np.uint8)       # last entry: every code point above the alphabet
        for i, c in enumerate(self.s["alphabet"]):
            self._lut[int(c)] = i
        po, pu = self.s["piece_u8_off"], self.s["piece_u8"]
        self.piece_surface = [pu[po[i]: po[i + 1]].tobytes().decode("utf-8") for i in range(1025)]
        self.surah = self.s["surah"]
        self.ayah = self.s["ayah"]

    def encode(self, text: str) -> np.ndarray:
        cp = np.frombuffer(text.encode("utf-32-le", "surrogatepass"), dtype="<u4")
        return self._lut[np.minimum(cp, len(self._lut) - 1)]

    def ids_to_text(self, ids) -> str:
        """SentencePiece decode_ids over the piece surfaces (leading whitespace markers of the
        pieces before the first emitted character are dropped; <unk> keeps its ' ⁇ ')."""
        out = ""
        for i in ids:
            surf = self.piece_surface[int(i)]
            if not out and int(i) != 0 and surf.startswith(" "):
                surf = surf[1:]
            out += surf
        return out

    def key_of(self, start: int, span: int):
        s, a = int(self.surah[start]), int(self.ayah[start])
        return (s, a, a + span - 1)

    def verse_index(self, surah: int, ayah: int) -> int:
        return int(self.s["surah_start"][surah - 1]) + ayah - 1

    def token_ids(self, start: int, span: int) -> np.ndarray:
        k = start * 6 + (span - 1)
        return self.s["tok"][self.s["tok_off"][k]: self.s["tok_off"][k + 1]]
