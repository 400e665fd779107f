"""Generate the golden fixtures that pin oracle/ against the reference.

Run ONLY in the build container (needs /root/reference):

    PYTHONHASHSEED=0 python tests/golden/gen_golden.py

It imports the unmodified reference post-logits code (ref_import.py) and writes

    tests/golden/normalizer_cases.json       shared/normalizer.py I/O pairs
    tests/golden/indel_known_answers.json    hand-computed + reference-shim LCS cases
    tests/golden/tokenizer_cases.json        SentencePiece decode/encode known answers
    tests/golden/retrieval_cases.json.gz     match_verse / search / pass-3 / candidate lists
    tests/golden/e2e_cases.json.gz           synthetic log-probs recipe -> greedy decode,
                                             per-candidate ctc_loss, ranking, predict() dict
    tests/golden/scoring_cases.json          runner.score_sequence known answers
    tests/golden/e2e_textweight_cases.json.gz  (section "textweight") the rerank and the decision with
                                             CTC_DIRECT_TEXT_WEIGHT = 0.35 / 2.0 on three of the e2e recipes

Fixtures are data (inputs + expected outputs); no reference source text is stored.
PYTHONHASHSEED is pinned because the reference's tie order follows set iteration
(SURVEY.md section 7 "Tie-breaking").
"""

from __future__ import annotations

import gzip
import json
import math
import os
import random
import sys
from pathlib import Path

if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

from ref_import import REF, indel_ratio, load_reference  # noqa: E402
from synth import synth_logits  # noqa: E402  (tests/synth.py: shared, reference-free)


def dump(name, obj, gz=False):
    txt = json.dumps(obj, ensure_ascii=False, separators=(",", ":"))
    path = HERE / name
    if gz:
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(txt.encode("utf-8"))
    else:
        path.write_text(txt, encoding="utf-8")
    print(f"wrote {name}: {path.stat().st_size} bytes")


def main():
    cd = load_reference()
    db = cd._db
    tok = cd._model.tokenizer
    rng = random.Random(20260630)
    sections = set(sys.argv[1:]) or {"small", "retrieval", "e2e"}
    if sections == {"textweight"}:
        return textweight_section(cd, tok, db)

    # ---------------- normalizer -------------------------------------------
    from shared.normalizer import normalize_arabic

    norm_inputs = []
    for idx in (0, 1, 6, 7, 261, 292, 1000, 2000, 3000, 4444, 6000, 6235):
        norm_inputs.append(db.verses[idx]["text_uthmani"])
    norm_inputs += [
        "",
        "   ",
        "﻿بِسْمِ  ٱللَّهِ‏ ٱلرَّحْمَٰنِ",
        "قَالَ، رَبِّ؟ ... إِنِّي!",
        "آمَنَ ٲ ٳ اٰ ٰ یے ک ـــ ﴿١٢٣﴾ ۝ ۞",
        "hello, world: 123 ١٢٣ ۱۲۳",
        " ⁇ ا ⁇ ",
    ]
    dump(
        "normalizer_cases.json",
        [{"in": s, "out": normalize_arabic(s)} for s in norm_inputs],
    )

    # ---------------- Indel ratio known answers ----------------------------
    hand = [
        # (a, b, lcs) -- LCS worked out by hand
        ("", "", 0),
        ("a", "", 0),
        ("abc", "abc", 3),
        ("abc", "abd", 2),
        ("kitten", "sitting", 4),  # i,t,t,n
        ("lewenstein", "levenshtein", 9),  # l e e n s t e i n
        ("abcdef", "fedcba", 1),
        ("aaaa", "aa", 2),
        ("abab", "baba", 3),
        ("x" * 70, "x" * 65 + "y" * 5, 65),  # crosses a 64-bit word
    ]
    ka = []
    for a, b, lcs in hand:
        r = indel_ratio(a, b)
        la, lb = len(a), len(b)
        expect = 1.0 if la + lb == 0 else 1.0 - (la + lb - 2 * lcs) / (la + lb)
        assert r == expect, (a, b, r, expect)
        ka.append({"a": a, "b": b, "lcs": lcs, "ratio": r, "hand": True})
    # published golden: retasy_005 -> 103:2 @ 0.9444 = 1 - 2/36
    assert round(1.0 - 2 / 36, 4) == 0.9444
    for _ in range(40):
        va = db.verses[rng.randrange(6236)]["text_clean"]
        vb = db.verses[rng.randrange(6236)]["text_clean_alt"]
        r = indel_ratio(va, vb)
        lcs = round((r * (len(va) + len(vb))) / 2)
        ka.append({"a": va, "b": vb, "lcs": lcs, "ratio": r, "hand": False})
    dump("indel_known_answers.json", ka)

    # ---------------- tokenizer --------------------------------------------
    tk = []
    for _ in range(40):
        n = rng.randrange(1, 40)
        ids = [rng.randrange(0, 1024) for _ in range(n)]
        tk.append({"ids": ids, "text": tok.ids_to_text(ids)})
    tk.append({"ids": [0], "text": tok.ids_to_text([0])})
    tk.append({"ids": [10, 10, 9], "text": tok.ids_to_text([10, 10, 9])})
    enc = []
    for idx in (0, 1, 7, 261, 6230, 6235):
        t = db.verses[idx]["text_clean"]
        enc.append({"text": t, "ids": tok.text_to_ids(t)})
    dump("tokenizer_cases.json", {"decode": tk, "encode": enc})

    # ---------------- retrieval cases ---------------------------------------
    def verse(s, a):
        return db.get_verse(s, a)

    def perturb(text, rate, seed):
        r = random.Random(seed)
        alphabet = "ابتثجحخدذرزسشصضطظعغفقكلمنهوي "
        out = []
        for ch in text:
            x = r.random()
            if x < rate / 3:
                continue
            if x < 2 * rate / 3:
                out.append(r.choice(alphabet))
                continue
            out.append(ch)
            if x < rate:
                out.append(r.choice(alphabet))
        return normalize_arabic("".join(out))

    def span_text(s, a0, a1):
        return " ".join(verse(s, a)["text_clean"] for a in range(a0, a1 + 1))

    transcripts = {
        "exact_1_1": verse(1, 1)["text_clean"],
        "exact_112_2": verse(112, 2)["text_clean"],
        "exact_36_1": verse(36, 1)["text_clean"],
        "exact_2_255": verse(2, 255)["text_clean"],
        "refrain_55_13_p": perturb(verse(55, 13)["text_clean"], 0.10, 1),
        "bsm_stripped_78_1": verse(78, 1)["text_clean_no_bsm"],
        "with_bsm_97_1": verse(97, 1)["text_clean"],
        "p05_103_2": perturb(verse(103, 2)["text_clean"], 0.05, 2),
        "p15_3_23": perturb(verse(3, 23)["text_clean"], 0.15, 3),
        "p30_59_23": perturb(verse(59, 23)["text_clean"], 0.30, 4),
        "p45_24_35": perturb(verse(24, 35)["text_clean"], 0.45, 5),
        "span_112_1_4": span_text(112, 1, 4),
        "span_103_1_3_p": perturb(span_text(103, 1, 3), 0.12, 6),
        "span_36_1_5_p": perturb(span_text(36, 1, 5), 0.2, 7),
        "span_114_1_6_p": perturb(span_text(114, 1, 6), 0.35, 8),
        "frag3_2_102": " ".join(verse(2, 102)["text_clean"].split()[5:8]),
        "frag6_2_282": " ".join(verse(2, 282)["text_clean"].split()[20:26]),
        "half_2_282_p": perturb(" ".join(verse(2, 282)["text_clean"].split()[:60]), 0.08, 9),
        "spaceless_67_1": verse(67, 1)["text_clean_no_bsm"].replace(" ", ""),
        "short_2w": "قل هو",
        "one_char": "ق",
        "two_char": "قل",
        "garbage_40": perturb("ا" * 40, 1.0, 10),
        "garbage_words": "زلط كبع شثق ضغظ خذج فقن ملك يوم",
        "unk_marks": "الحمد ⁇ لله رب ⁇ العالمين",
    }

    def keyfloat(x):
        return float(x)

    ret_cases = []
    for name, t in (transcripts.items() if "retrieval" in sections else ()):
        base = db.match_verse(
            t, threshold=0.0, max_span=6, return_top_k=100, use_trigram_index=True
        )
        tri = db._trigram_candidates(normalize_arabic(t), top_k=50)
        srch = db.search(t, top_k=100)
        norm = t
        spaceless = norm.replace(" ", "")
        scored = []
        for v in db.verses:
            s = max(
                indel_ratio(norm, v["text_clean"]),
                indel_ratio(spaceless, v["text_clean"].replace(" ", "")),
            )
            scored.append((s, v))
        p3 = sorted(scored, key=lambda x: x[0], reverse=True)[:100]
        cands, base2 = cd._build_candidates(t)
        case = {
            "name": name,
            "transcript": t,
            "trigram_top50": [int(i) for i in tri],
            "match": None
            if base is None
            else {
                "surah": base["surah"],
                "ayah": base["ayah"],
                "ayah_end": base.get("ayah_end"),
                "score": keyfloat(base["score"]),
                "raw_score": keyfloat(base["raw_score"]),
                "runners_up": [
                    [r["surah"], r["ayah"], keyfloat(r["score"])] for r in base["runners_up"]
                ],
            },
            "search100": [[v["surah"], v["ayah"], keyfloat(v["score"])] for v in srch],
            "pass3_100": [[v["surah"], v["ayah"], keyfloat(s)] for s, v in p3],
            "candidates": [[c["surah"], c["ayah"], c["ayah_end"]] for c in cands],
            "cand_scores": [keyfloat(c.get("score") or 0.0) for c in cands],
        }
        ret_cases.append(case)
        print(
            name,
            len(t),
            None if base is None else (base["surah"], base["ayah"], base.get("ayah_end"), round(base["score"], 4)),
            len(cands),
        )
    if "retrieval" in sections:
        dump("retrieval_cases.json.gz", ret_cases, gz=True)

    # ---------------- end-to-end: logits recipe -> predict ---------------------
    import torch  # noqa: F401

    def run_e2e(name, ids_path, T, seed, noise, boost, rep):
        logits = synth_logits(ids_path, T, seed=seed, noise=noise, boost=boost, rep=rep)
        lp = torch.log_softmax(torch.from_numpy(logits), dim=-1).numpy()
        transcript = cd._greedy_decode(lp)
        am = lp.argmax(-1)
        dedup, prev = [], -1
        for i in am:
            i = int(i)
            if i != prev and i != 1024:
                dedup.append(i)
            prev = i
        out = {
            "name": name,
            "recipe": {"ids": ids_path, "T": T, "seed": seed, "noise": noise, "boost": boost, "rep": rep},
            "greedy_ids": dedup,
            "transcript": transcript,
        }
        if not transcript.strip():
            out["result"] = {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0, "source": None}
            return out
        cands, base = cd._build_candidates(transcript)
        use_ctc = base is None or float(base.get("score", 0.0)) < cd.FALLBACK_THRESHOLD
        # always record the rerank so the kernel is pinned on gate-pass cases too
        ranked = cd._ctc_rerank(lp, cands)
        out["base"] = None if base is None else [base["surah"], base["ayah"], base.get("ayah_end"), float(base["score"])]
        out["use_ctc"] = bool(use_ctc)
        out["n_candidates"] = len(cands)
        out["cand_keys"] = [[c["surah"], c["ayah"], c["ayah_end"]] for c in cands]
        out["ctc_loss"] = [
            (float(c["ctc_loss"]) if math.isfinite(c["ctc_loss"]) else None) for c in cands
        ]
        out["ctc_len"] = [int(c["ctc_len"]) for c in cands]
        out["ranked_keys"] = [[c["surah"], c["ayah"], c["ayah_end"]] for c in ranked[:20]]
        out["ranked_final"] = [float(c["final_score"]) for c in ranked[:20]]
        if use_ctc and ranked:
            best = ranked[0]
            score = math.exp(-best["ctc_norm_loss"]) if math.isfinite(best["ctc_norm_loss"]) else 0.0
            src = "ctc"
        elif base:
            best, score, src = base, float(base.get("score", 0.0)), "text"
        else:
            best = None
        if best is None:
            out["result"] = {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0, "source": None}
        else:
            out["result"] = {
                "surah": best["surah"],
                "ayah": best["ayah"],
                "ayah_end": best.get("ayah_end") or best["ayah"],
                "score": round(score, 4),
                "score_raw": float(score),
                "source": src,
            }
        print(name, T, len(dedup), out["result"], "use_ctc", use_ctc, "ncand", len(cands))
        return out

    def ids_of(text):
        return [int(i) for i in tok.text_to_ids(text)]

    def corrupt_ids(ids, rate, seed):
        r = random.Random(seed)
        out = []
        for i in ids:
            x = r.random()
            if x < rate / 2:
                continue
            if x < rate:
                out.append(r.randrange(1, 1024))
            else:
                out.append(i)
        return out

    e2e = []
    if "e2e" not in sections:
        return
    e2e.append(run_e2e("clean_1_1", ids_of(verse(1, 1)["text_clean"]), 40, 1, 1.0, 8.0, 2))
    e2e.append(run_e2e("clean_112_2", ids_of(verse(112, 2)["text_clean"]), 20, 2, 1.0, 8.0, 2))
    e2e.append(run_e2e("corrupt_103_2", corrupt_ids(ids_of(verse(103, 2)["text_clean"]), 0.35, 3), 48, 3, 1.0, 7.0, 2))
    e2e.append(run_e2e("corrupt_114_1_6", corrupt_ids(ids_of(span_text(114, 1, 6)), 0.3, 4), 150, 4, 1.0, 6.0, 2))
    e2e.append(run_e2e("corrupt_36_1_5", corrupt_ids(ids_of(span_text(36, 1, 5)), 0.35, 5), 126, 5, 1.0, 6.0, 3))
    e2e.append(run_e2e("corrupt_55_1_4", corrupt_ids(ids_of(span_text(55, 1, 4)), 0.4, 6), 90, 6, 1.0, 6.0, 3))
    e2e.append(run_e2e("noise_only", [], 63, 7, 1.0, 0.0, 1))
    e2e.append(run_e2e("all_blank", [], 30, 8, 0.0, 0.0, 1))
    e2e.append(run_e2e("long_2_255", corrupt_ids(ids_of(verse(2, 255)["text_clean"]), 0.3, 9), 251, 9, 1.0, 6.0, 2))
    e2e.append(run_e2e("tight_T_103_1", ids_of(verse(103, 1)["text_clean_no_bsm"]), 2 * len(ids_of(verse(103, 1)["text_clean_no_bsm"])) + 1, 10, 1.0, 8.0, 1))
    dump("e2e_cases.json.gz", e2e, gz=True)


def textweight_section(cd, tok, db):
    """CTC_DIRECT_TEXT_WEIGHT != 0 (c2c-direct/run.py:62-74,371-373): final = -norm_loss + TEXT_WEIGHT * text_score -
    penalty, where text_score is whatever `score` the candidate carries (match_verse's unrounded best, the runners-up
    rounded to 3 dp, search()'s fragment scores, pass 3's ratios, 0.0 for expanded spans)."""
    import torch

    def verse(s, a):
        return db.get_verse(s, a)

    def span_text(s, a0, a1):
        chunk = [verse(s, a) for a in range(a0, a1 + 1)]
        first = chunk[0].get("text_clean_no_bsm") or chunk[0]["text_clean"]
        return " ".join([first] + [v["text_clean"] for v in chunk[1:]])

    def ids_of(text):
        return [int(i) for i in tok.text_to_ids(text)]

    def corrupt_ids(ids, rate, seed):   # (same recipe as the e2e section)
        r = random.Random(seed)
        out = []
        for i in ids:
            x = r.random()
            if x < rate / 2:
                continue
            if x < rate:
                out.append(r.randrange(1, 1024))
                continue
            out.append(i)
        return out

    recipes = [
        ("corrupt_103_2", corrupt_ids(ids_of(verse(103, 2)["text_clean"]), 0.35, 3), 48, 3, 1.0, 7.0, 2),
        ("corrupt_36_1_5", corrupt_ids(ids_of(span_text(36, 1, 5)), 0.35, 5), 126, 5, 1.0, 6.0, 3),
        ("corrupt_55_1_4", corrupt_ids(ids_of(span_text(55, 1, 4)), 0.4, 6), 90, 6, 1.0, 6.0, 3),
    ]
    out = []
    for name, ids_path, T, seed, noise, boost, rep in recipes:
        logits = synth_logits(ids_path, T, seed=seed, noise=noise, boost=boost, rep=rep)
        lp = torch.log_softmax(torch.from_numpy(logits), dim=-1).numpy()
        transcript = cd._greedy_decode(lp)
        cands, base = cd._build_candidates(transcript)
        use_ctc = base is None or float(base.get("score", 0.0)) < cd.FALLBACK_THRESHOLD
        for tw in (0.35, 2.0):
            keep = cd.TEXT_WEIGHT
            cd.TEXT_WEIGHT = tw
            try:
                ranked = cd._ctc_rerank(lp, [dict(c) for c in cands])
            finally:
                cd.TEXT_WEIGHT = keep
            best = ranked[0]
            out.append({
                "name": name, "text_weight": tw,
                "recipe": {"ids": ids_path, "T": T, "seed": seed, "noise": noise, "boost": boost, "rep": rep},
                "transcript": transcript, "use_ctc": bool(use_ctc), "n_candidates": len(cands),
                "ranked_keys": [[c["surah"], c["ayah"], c["ayah_end"]] for c in ranked[:20]],
                "ranked_final": [float(c["final_score"]) for c in ranked[:20]],
                "ranked_text_score": [float(c.get("score") or 0.0) for c in ranked[:20]],
                "winner": [best["surah"], best["ayah"], best["ayah_end"]],
                "winner_score_raw": float(math.exp(-best["ctc_norm_loss"])),
            })
            print(name, tw, out[-1]["winner"], out[-1]["ranked_final"][:3])
    dump("e2e_textweight_cases.json.gz", out, gz=True)

    # ---------------- score_sequence known answers -----------------------------
    # inputs are the six hand cases of tests/test_scoring.py:8-59 (data), outputs
    # computed by the reference's own function.
    import importlib

    sys.modules.setdefault("shared.streaming", importlib.import_module("types").ModuleType("shared.streaming"))
    sys.modules["shared.streaming"].StreamingPipeline = object
    runner_spec = importlib.util.spec_from_file_location("_ref_runner", str(REF / "benchmark/runner.py"))
    runner = importlib.util.module_from_spec(runner_spec)
    runner_spec.loader.exec_module(runner)
    E = lambda *p: [{"surah": s, "ayah": a} for s, a in p]  # noqa: E731
    sc_in = [
        (E((103, 1), (103, 2), (103, 3)), E((103, 1), (103, 2), (103, 3))),
        (E((103, 1), (103, 2), (103, 3)), E((103, 1), (103, 3))),
        (E((1, 1)), E((2, 1))),
        (E((2, 255)), E((2, 255))),
        (E((1, 1)), []),
        (E((1, 1)), E((1, 1), (1, 2))),
        ([], E((1, 1))),
        (E((1, 1), (1, 2)), E((1, 2), (1, 1))),
    ]
    sc = [{"expected": e, "predicted": p, "out": runner.score_sequence(e, p)} for e, p in sc_in]
    em_in = [
        {"surah": 112, "ayah": 2, "ayah_end": 3, "score": 0.9811},
        {"surah": 1, "ayah": 1, "ayah_end": None, "score": 1.0},
        {"surah": 0, "ayah": 0, "ayah_end": None, "score": 0.0},
        {},
    ]
    em = [{"in": r, "out": runner._predict_to_emissions(r)} for r in em_in]
    dump("scoring_cases.json", {"score_sequence": sc, "emissions": em})


if __name__ == "__main__":
    main()
