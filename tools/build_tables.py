"""Build offline-tarteel_amd/data/qverse_tables.bin -- the static tables of the hot path.

Build-container tool (needs the reference's *data* files, not its code):

    python tools/build_tables.py [--quran /root/reference/data/quran.json]
                                 [--tokenizer /root/reference/web/frontend/public/tokenizer.model]

What the reference computes at process start (QuranDB.__init__ + _build_trigram_index,
shared/quran_db.py:39-65,151-171) and lazily per candidate (_token_ids,
experiments/c2c-direct/run.py:215-221) becomes one flat, pointer-free blob that the
HIP library uploads to HBM once:

  alphabet            u32[K]    code points; code 0 is ' ', code 63 = "matches nothing"
  surah/ayah          u8[N]/u16[N], surah_start i32[115], surah_len i32[114]
  clean/alt/nobsm     u8 code strings + u32 offsets  (text_clean, text_clean_alt,
                      text_clean_no_bsm -- quran_db.py:45-59)
  *_nw                u16[N]    word counts (for _fragment_score, :221-231)
  tok / tok_off       u16 ids + u32[N*6+1]: SentencePiece ids of the CTC text of every
                      (start verse, span length 1..6) -- singles use text_clean, spans use
                      " ".join(first.no_bsm or first.clean, rest.clean) tokenised AS JOINED
                      TEXT (c2c-direct/run.py:224-248)
  piece_*             per vocabulary id: normalised code string (what the id contributes to
                      the normalised transcript) and raw UTF-8 surface (host-side text)
  tri_keys/tri_idf    sorted packed trigrams (c0<<12|c1<<6|c2) and ln(N/df) (:164-171)
  vtri / vtri_off     forward index: sorted trigram ids present in each verse (union over its
                      three texts, :156-163)

File layout: "QVTB0001", u32 n_sections, u32 0, n x {char name[24]; u64 offset; u64 nbytes},
then 64-byte aligned payloads.
"""

from __future__ import annotations

import argparse
import json
import math
import struct
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import offline_tarteel_amd  # noqa: E402
from offline_tarteel_amd.normalizer import normalize_arabic  # noqa: E402

MAX_SPAN = 6
OTHER = 63
BSM = normalize_arabic("بسم الله الرحمن الرحيم")


def write_blob(path: Path, sections: dict[str, np.ndarray]) -> None:
    names = list(sections)
    hdr = 16 + 40 * len(names)
    off = (hdr + 63) // 64 * 64
    table = []
    for n in names:
        a = np.ascontiguousarray(sections[n])
        table.append((n, off, a.nbytes))
        off = (off + a.nbytes + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(b"QVTB0001")
        f.write(struct.pack("<II", len(names), 0))
        for n, o, nb in table:
            f.write(n.encode("ascii").ljust(24, b"\0"))
            f.write(struct.pack("<QQ", o, nb))
        for n, o, nb in table:
            f.seek(o)
            f.write(np.ascontiguousarray(sections[n]).tobytes())
        f.seek(off - 1) if off > f.tell() else None
        if off > f.tell():
            f.write(b"\0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quran", default="/root/reference/data/quran.json")
    ap.add_argument("--tokenizer", default="/root/reference/web/frontend/public/tokenizer.model")
    ap.add_argument("--out", default=str(offline_tarteel_amd.TABLES_PATH))
    ap.add_argument("--emit-json", default="",
                    help='also write the upstream token-table artefact quran_ctc_tokens.json: "surah:ayah:ayah_end" -> ids')
    ap.add_argument("--json-max-span", type=int, default=6,
                    help="ayat per span in the JSON.  6 reproduces the upstream counts exactly (35,717 keys, 29,481 "
                         "multi-ayah spans, PLAN.md:102) and equals CTC_DIRECT_MAX_SPAN; PLAN.md:121-124's prose says "
                         "'max span length 5', which would give 30,043 keys")
    ap.add_argument("--json-span-text", choices=("plan", "hot-path"), default="plan",
                    help='plan: spans are " ".join(text_clean) as PLAN.md:102 specifies the upstream file; '
                         "hot-path: the first ayah loses its bismillah like c2c-direct/run.py:235-248 (= the binary table)")
    args = ap.parse_args()

    import sentencepiece as spm

    sp = spm.SentencePieceProcessor(model_file=args.tokenizer)
    verses = json.load(open(args.quran, encoding="utf-8"))
    N = len(verses)

    clean, alt, nobsm = [], [], []
    for v in verses:
        c = v["text_clean"].lstrip("﻿")
        a = normalize_arabic(v["text_uthmani"]).lstrip("﻿")
        nb = ""
        if v["ayah"] == 1 and v["surah"] not in (1, 9) and c.startswith(BSM):
            nb = c[len(BSM):].strip()
        clean.append(c)
        alt.append(a)
        nobsm.append(nb)

    chars = sorted(set("".join(clean)) | set("".join(alt)) | set("".join(nobsm)))
    assert chars[0] == " " and len(chars) < OTHER, (chars[:3], len(chars))
    code_of = {ch: i for i, ch in enumerate(chars)}
    alphabet = np.array([ord(c) for c in chars], dtype=np.uint32)

    def enc(s: str) -> np.ndarray:
        return np.array([code_of.get(ch, OTHER) for ch in s], dtype=np.uint8)

    def pack(strs):
        off = np.zeros(len(strs) + 1, dtype=np.uint32)
        parts = []
        for i, s in enumerate(strs):
            e = enc(s)
            parts.append(e)
            off[i + 1] = off[i] + len(e)
        return off, (np.concatenate(parts) if parts else np.zeros(0, np.uint8))

    clean_off, clean_codes = pack(clean)
    alt_off, alt_codes = pack(alt)
    nobsm_off, nobsm_codes = pack(nobsm)
    nw = lambda strs: np.array([len(s.split()) for s in strs], dtype=np.uint16)  # noqa: E731

    surah = np.array([v["surah"] for v in verses], dtype=np.uint8)
    ayah = np.array([v["ayah"] for v in verses], dtype=np.uint16)
    n_surah = int(surah.max())
    surah_start = np.zeros(n_surah + 1, dtype=np.int32)
    surah_len = np.zeros(n_surah, dtype=np.int32)
    for s in range(1, n_surah + 1):
        idx = np.nonzero(surah == s)[0]
        assert (np.diff(idx) == 1).all() and (ayah[idx] == np.arange(1, len(idx) + 1)).all()
        surah_start[s - 1] = idx[0]
        surah_len[s - 1] = len(idx)
    surah_start[n_surah] = N

    # ---- CTC token table --------------------------------------------------------
    tok_off = np.zeros(N * MAX_SPAN + 1, dtype=np.uint32)
    toks = []
    total = 0
    for i in range(N):
        s = int(surah[i])
        last = surah_start[s - 1] + surah_len[s - 1] - 1
        for k in range(1, MAX_SPAN + 1):
            ids = []
            if i + k - 1 <= last:
                if k == 1:
                    text = clean[i]
                else:
                    text = " ".join([nobsm[i] or clean[i]] + [clean[j] for j in range(i + 1, i + k)])
                ids = sp.encode_as_ids(text)
                assert all(0 <= t < 1024 for t in ids) and ids
            toks.extend(ids)
            total += len(ids)
            tok_off[i * MAX_SPAN + k] = total
    tok = np.array(toks, dtype=np.uint16)

    if args.emit_json:
        # The browser runtime's precomputed table (web/frontend/src/worker/quran-text-adapter.ts:16-31,
        # lib/quran-db.ts:718-719; upstream file quran_ctc_tokens.json, a missing blob in the reference tree):
        # key "surah:ayah:ayah_end" (single verses: ayah_end = ayah) -> SentencePiece ids of the text tokenised
        # ONCE as joined text, spans of at most --json-max-span ayat inside one surah (PLAN.md:102-103,121-124).
        table = {}
        for i in range(N):
            s_no, a_no = int(surah[i]), int(ayah[i])
            last = surah_start[s_no - 1] + surah_len[s_no - 1] - 1
            for k in range(1, args.json_max_span + 1):
                if i + k - 1 > last:
                    break
                if k == 1:
                    text = clean[i]
                elif args.json_span_text == "plan":
                    text = " ".join(clean[j] for j in range(i, i + k))
                else:
                    text = " ".join([nobsm[i] or clean[i]] + [clean[j] for j in range(i + 1, i + k)])
                table[f"{s_no}:{a_no}:{a_no + k - 1}"] = [int(t) for t in sp.encode_as_ids(text)]
        n_multi = sum(1 for key in table if key.split(":")[1] != key.split(":")[2])
        Path(args.emit_json).write_text(json.dumps(table, separators=(",", ":")), encoding="utf-8")
        print(f"wrote {args.emit_json}: {len(table)} keys ({n_multi} multi-ayah spans, max span {args.json_max_span}, "
              f"span text: {args.json_span_text}), {Path(args.emit_json).stat().st_size} B")

    # ---- vocabulary pieces -------------------------------------------------------
    V = sp.get_piece_size()
    anchor = sp.piece_to_id("ا")
    assert anchor > 0
    piece_off = np.zeros(V + 2, dtype=np.uint32)
    piece_u8_off = np.zeros(V + 2, dtype=np.uint32)
    pc, pu = [], []
    for i in range(V + 1):
        if i < V:
            surf = sp.decode_ids([anchor, i, anchor])[1:-1]  # surface incl. its leading space
        else:
            surf = ""  # blank (1024)
        # per-character normalisation is context-free on this vocabulary: no U+0670, so the
        # only cross-character rule (alef + superscript alef) can never fire across pieces
        assert "ٰ" not in surf
        kept = []
        for ch in surf:
            if ch.isspace():
                kept.append(" ")
                continue
            n = normalize_arabic(ch, collapse_whitespace=False)
            kept.extend(n)
        codes = enc("".join(kept))
        pc.append(codes)
        piece_off[i + 1] = piece_off[i] + len(codes)
        b = np.frombuffer(surf.encode("utf-8"), dtype=np.uint8)
        pu.append(b)
        piece_u8_off[i + 1] = piece_u8_off[i] + len(b)
    piece_codes = np.concatenate(pc)
    piece_u8 = np.concatenate(pu)

    # ---- trigram index -------------------------------------------------------------
    def tris(e: np.ndarray) -> set[int]:
        if len(e) < 3:
            return set()
        e = e.astype(np.uint32)
        return set(((e[:-2] << 12) | (e[1:-1] << 6) | e[2:]).tolist())

    per_verse = []
    df: dict[int, int] = {}
    for i in range(N):
        t = tris(enc(clean[i])) | tris(enc(alt[i]))
        if nobsm[i]:
            t |= tris(enc(nobsm[i]))
        per_verse.append(t)
        for k in t:
            df[k] = df.get(k, 0) + 1
    keys = np.array(sorted(df), dtype=np.uint32)
    idf = np.array([math.log(N / df[int(k)]) for k in keys], dtype=np.float64)
    kid = {int(k): j for j, k in enumerate(keys)}
    vtri_off = np.zeros(N + 1, dtype=np.uint32)
    vt = []
    for i, t in enumerate(per_verse):
        ids = sorted(kid[k] for k in t)
        vt.extend(ids)
        vtri_off[i + 1] = vtri_off[i] + len(ids)
    vtri = np.array(vt, dtype=np.uint16)
    assert len(keys) < 65535

    meta = np.zeros(16, dtype=np.int32)
    meta[:8] = [N, n_surah, len(chars), MAX_SPAN, V + 1, V, len(keys), OTHER]

    sections = {
        "meta": meta,
        "alphabet": alphabet,
        "surah": surah,
        "ayah": ayah,
        "surah_start": surah_start,
        "surah_len": surah_len,
        "clean_off": clean_off,
        "clean": clean_codes,
        "alt_off": alt_off,
        "alt": alt_codes,
        "nobsm_off": nobsm_off,
        "nobsm": nobsm_codes,
        "clean_nw": nw(clean),
        "alt_nw": nw(alt),
        "nobsm_nw": nw(nobsm),
        "tok_off": tok_off,
        "tok": tok,
        "piece_off": piece_off,
        "piece_codes": piece_codes,
        "piece_u8_off": piece_u8_off,
        "piece_u8": piece_u8,
        "tri_keys": keys,
        "tri_idf": idf,
        "vtri_off": vtri_off,
        "vtri": vtri,
    }
    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    write_blob(out, sections)
    print(
        f"wrote {out} ({out.stat().st_size} B): N={N} K={len(chars)} tok={len(tok)} "
        f"tri={len(keys)} postings={len(vtri)} pieces={V}"
    )


if __name__ == "__main__":
    main()
