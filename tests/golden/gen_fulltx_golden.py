"""Golden fixtures for StreamingPipeline.run_on_full_transcript (shared/streaming.py:58-105) and
the match_verse variant it calls (shared/quran_db.py:244-371 with use_trigram_index=False,
max_span=8 and a continuation hint: _continuation_bonuses :121-146, _suffix_prefix_score :188-208).

Run ONLY in the build container (needs /root/reference; slow: the reference scans all 6,236
verses with a pure-Python Indel ratio per call):

    PYTHONHASHSEED=0 python tests/golden/gen_fulltx_golden.py

Writes tests/golden/fulltx_cases.json.gz:
    match    match_verse(text, max_span=8, hint=h) -> None | surah, ayah, ayah_end, score,
             raw_score, bonus, n_words(text_clean)          (threshold 0.3, the default)
    full     run_on_full_transcript("x.wav", lambda p: text) -> emissions
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""

from __future__ import annotations

import gzip
import json
import os
import random
import sys
from pathlib import Path

if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

from gen_tracker_golden import corrupt  # noqa: E402
from ref_import import load_reference  # noqa: E402


def main():
    cd = load_reference()
    db = cd._db
    from shared import streaming as st

    rng = random.Random(20260701)
    by_ref = {(v["surah"], v["ayah"]): v for v in db.verses}

    def words(ref, a=None, b=None):
        return " ".join(by_ref[ref]["text_clean"].split()[a:b])

    match_in = [
        (words((112, 1)), None), (words((112, 2)), (112, 1)), (words((112, 3)) + " " + words((112, 4)), (112, 2)),
        (words((2, 1)), (1, 7)), (words((2, 2)), (1, 7)), (words((114, 6)), (114, 5)), (words((1, 1)), (114, 6)),
        (words((55, 13)), (55, 12)), (words((55, 13)), None), (words((55, 13)), (55, 20)),
        (words((103, 1), -1) + " " + words((103, 2)), (103, 1)),           # residual word of the previous verse
        (words((36, 2), -2) + " " + words((36, 3)) + " " + words((36, 4), 0, 2), (36, 2)),
        (words((2, 255), 0, 12), None), (words((1, 2)) + " " + words((1, 3)) + " " + words((1, 4)), (1, 1)),
        (" ".join(words((94, a)) for a in range(1, 9)), None),               # whole surah: span of 8
        (" ".join(words((109, a)) for a in range(1, 7)), None),
        (corrupt(rng, words((67, 1)) + " " + words((67, 2)), 0.1), None),
        (corrupt(rng, words((96, 1)), 0.2), (95, 8)), ("xyz abc", None), ("الله", (2, 254)),
    ]
    match = []
    for text, hint in match_in:
        r = db.match_verse(text, max_span=8, hint=tuple(hint) if hint else None)
        match.append({"text": text, "hint": list(hint) if hint else None,
                      "result": None if r is None else {
                          "surah": r["surah"], "ayah": r["ayah"], "ayah_end": r.get("ayah_end"), "score": r["score"],
                          "raw_score": r["raw_score"], "bonus": r["bonus"], "n_words": len(r["text_clean"].split())}})
        print("match", len(match), flush=True)

    pipe = st.StreamingPipeline(db)
    full_in = [words((112, 1)), " ".join(words((112, a)) for a in range(1, 5)),
               " ".join(words((103, a)) for a in range(1, 4)), " ".join(words((1, a)) for a in range(1, 8)),
               words((2, 255)), corrupt(rng, " ".join(words((93, a)) for a in range(1, 6)), 0.08),
               words((113, 5)) + " " + words((114, 1)) + " " + words((114, 2)), "", "xyz abc def"]
    full = []
    for text in full_in:
        full.append({"text": text, "emissions": pipe.run_on_full_transcript("x.wav", lambda p, t=text: t)})
        print("full", len(full), flush=True)

    path = HERE / "fulltx_cases.json.gz"
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps({"match": match, "full": full}, ensure_ascii=False, separators=(",", ":")).encode("utf-8"))
    print(f"wrote {path.name}: {path.stat().st_size} bytes")


if __name__ == "__main__":
    main()
