#!/usr/bin/env python
"""Differential fuzz of the span pass: k_spans2 (prefix-shared, the default) against k_spans (one walk per span) on random
corrupted multi-ayah recitations from everywhere in the text -- weighted towards the surahs of very short ayat, whose
two-chunk blocks are the case the first version of k_spans2 got wrong -- through the hot path's retrieval (max_span 6) and
through qv_match_verse (max_span 8, with and without a continuation hint).     python tools/fuzz_spans.py [--n 3000] [--seed 1]"""
import argparse
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import offline_tarteel_amd  # noqa: F401
    from offline_tarteel_amd.engine import Engine
    from oracle.oracle import Oracle, normalize_arabic

    orc = Oracle()
    eng = Engine(device=0, with_model=False, max_batch=16)
    rnd = random.Random(a.seed)
    n_verses = len(orc.surah)
    letters = [ch for ch in orc.alphabet if ch != " "][:28]
    short_surahs = [26, 37, 44, 52, 53, 54, 55, 56, 69, 70, 74, 75, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93]
    bad = 0
    for it in range(a.n):
        if rnd.random() < 0.5:
            s = rnd.choice(short_surahs)
            v0 = int(orc.t["surah_start"][s - 1]) + rnd.randrange(int(orc.t["surah_len"][s - 1]))
        else:
            v0 = rnd.randrange(n_verses)
        k = rnd.randrange(1, 12)
        last = int(orc.t["surah_start"][orc.surah[v0]]) - 1
        words = []
        for d in range(k):
            if v0 + d > last:
                break
            words += orc.verse_text(v0 + d).split()
        full = " ".join(words)
        rate = rnd.choice((0.0, 0.05, 0.15, 0.4))
        out = []
        for ch in full:
            x = rnd.random()
            if x < rate / 3:
                continue
            out.append(rnd.choice(letters) if x < 2 * rate / 3 else ch)
        cut = rnd.choice((24, 60, 100, 128, 129, 200, 256, 257, 400, 512, 513, 700, 1024))
        t = normalize_arabic(" ".join("".join(out).split())[:cut]).strip()
        if len(t) < 8:
            continue
        hint = None
        if rnd.random() < 0.3:
            hv = max(0, v0 - 1)
            hint = (int(orc.surah[hv]), int(orc.t["ayah"][hv]))
        got = {}
        for var in (0, 1):
            eng.kernel_variant(2, var)
            r = eng.debug_retrieve(t)
            m = eng.match_verse(t, threshold=0.0, max_span=8, hint=hint)
            got[var] = ((r["base_start"], r["base_span"], r["base_score"], r["cand_start"].tolist(), r["cand_span"].tolist(), r["cand_score"].tolist()), m)
        if got[0] != got[1]:
            bad += 1
            print("MISMATCH", it, len(t), v0, k, rate, got[0][0][:3], got[1][0][:3], got[0][1], got[1][1])
    eng.kernel_variant(2, -1)
    eng.close()
    print(f"fuzz_spans: {a.n} texts, {bad} mismatches between k_spans and k_spans2")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
