"""Generate the golden fixtures for the streaming row:
shared/verse_tracker.py::VerseTracker and shared/streaming.py::StreamingPipeline.

Run ONLY in the build container (needs /root/reference):

    PYTHONHASHSEED=0 python tests/golden/gen_tracker_golden.py

Writes tests/golden/tracker_cases.json.gz with three sections, all produced by the
*unmodified* reference classes (imported through ref_import.py, i.e. with the Indel-ratio
stand-in for the absent Levenshtein package):

    best_match   VerseTracker._find_best_match(text) for full verses, word prefixes, verse
                 + start of the next one, corrupted text, bismillah variants, repeated
                 verses with and without a last emission (continuation bonus), both modes
    run_on_text  StreamingPipeline.run_on_text(snapshots) emission lists
    chunked      StreamingPipeline.run_on_audio_chunked with a scripted transcribe_fn
                 (str and {"text", "avg_logprob"} returns) over a synthetic sample count:
                 pins the chunk walk, the confidence gate and the tentative/confirmed logic.
                 The reference writes each chunk to a temporary WAV through soundfile, which
                 is absent here; the generator replaces that write with a no-op, so the
                 scripted transcripts are what the pipeline sees.

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

import numpy as np  # noqa: E402

from ref_import import load_reference  # noqa: E402


def corrupt(rng, text, rate):
    alphabet = sorted(set(text.replace(" ", "")))
    out = []
    for ch in text:
        r = rng.random()
        if ch != " " and r < rate:
            k = rng.randrange(3)
            if k == 0:
                continue                      # deletion
            if k == 1:
                out.append(rng.choice(alphabet))  # substitution
                continue
            out.append(ch)
            out.append(rng.choice(alphabet))  # insertion
            continue
        out.append(ch)
    return " ".join("".join(out).split())


def main():
    cd = load_reference()
    db = cd._db
    from shared import streaming as st
    from shared.verse_tracker import VerseTracker

    rng = random.Random(20260630)
    by_ref = {(v["surah"], v["ayah"]): v for v in db.verses}

    def words(ref, a=None, b=None):
        return " ".join(by_ref[ref]["text_clean"].split()[a:b])

    # ------------------------------------------------------------------ best_match ----
    texts = []
    refs = [(1, 1), (1, 2), (1, 7), (2, 1), (2, 2), (2, 255), (2, 282), (36, 1), (36, 2), (55, 13), (55, 16),
            (77, 15), (94, 5), (94, 6), (103, 1), (103, 3), (109, 3), (109, 5), (112, 1), (112, 4), (114, 6),
            (96, 1), (97, 1), (9, 1), (27, 30)]
    for r in refs:
        texts.append((words(r), None))
    for r in [(2, 255), (2, 282), (1, 7), (103, 3), (36, 12), (18, 10)]:
        n = len(by_ref[r]["text_clean"].split())
        for k in sorted({1, 2, 3, max(1, n // 2), max(1, (4 * n) // 5), max(1, n - 1)}):
            texts.append((words(r, 0, k), None))
    for r, nxt in [((112, 1), (112, 2)), ((1, 2), (1, 3)), ((103, 1), (103, 2)), ((36, 1), (36, 2)), ((2, 255), (2, 256)),
                   ((113, 5), (114, 1)), ((55, 12), (55, 13))]:
        for k in (1, 3):
            texts.append((words(r) + " " + words(nxt, 0, k), None))
            texts.append((words(r) + " " + words(nxt, 0, k), r))
    for r in [(1, 1), (2, 1), (96, 1), (97, 1), (112, 1)]:      # first ayat: bismillah and no_bsm variants
        v = by_ref[r]
        if v.get("text_clean_no_bsm"):
            texts.append((v["text_clean_no_bsm"], None))
            texts.append((v["text_clean_no_bsm"], (r[0] - 1, len(db._by_surah[r[0] - 1])) if r[0] > 1 else None))
    for r in [(55, 13), (55, 16), (55, 18), (77, 15), (77, 19), (94, 5), (94, 6), (109, 3), (109, 5), (26, 9), (26, 68)]:
        prev = (r[0], r[1] - 1)
        texts.append((words(r), prev))                          # repeated verses: continuation decides
        texts.append((words(r), (r[0], r[1] + 5) if (r[0], r[1] + 5) in by_ref else None))
    for _ in range(40):
        v = db.verses[rng.randrange(len(db.verses))]
        t = corrupt(rng, v["text_clean"], rng.choice((0.05, 0.15, 0.3)))
        last = None
        if rng.random() < 0.4:
            last = (v["surah"], v["ayah"] - 1) if v["ayah"] > 1 else None
        if t.strip():
            texts.append((t, last))
    for _ in range(12):                                          # two or three consecutive verses glued
        i = rng.randrange(len(db.verses) - 3)
        k = rng.choice((2, 3))
        t = " ".join(db.verses[i + j]["text_clean"] for j in range(k))
        if len(t) <= 900:
            texts.append((corrupt(rng, t, 0.1), None))
    texts += [("xyz abc", None), ("قل", None), ("الله", None), ("الحمد", None), ("قل هو", None),
              ("hello الله world", None), ("ا", None)]

    best = []
    for mode in (False, True):
        for text, last in texts:
            tr = VerseTracker(db, last_emission=tuple(last) if last else None, streaming_mode=mode)
            m = tr._find_best_match(text)
            best.append({
                "text": text, "last": list(last) if last else None, "streaming": mode,
                "match": None if m is None else {
                    "surah": m["surah"], "ayah": m["ayah"], "n_words": len(m["text_clean"].split()),
                    "score": m["score"]},
            })

    # ------------------------------------------------------------------ run_on_text ----
    pipe = st.StreamingPipeline(db)
    seqs = [[(112, a) for a in range(1, 5)], [(1, a) for a in range(1, 8)], [(103, a) for a in range(1, 4)],
            [(36, a) for a in range(1, 6)], [(55, a) for a in range(10, 17)], [(94, a) for a in range(1, 9)],
            [(113, 4), (113, 5), (114, 1), (114, 2)], [(2, 255)], [(109, a) for a in range(1, 7)]]
    run_text = []
    for seq in seqs:
        allw = " ".join(by_ref[r]["text_clean"] for r in seq).split()
        for step, rate in ((1, 0.0), (3, 0.0), (4, 0.08), (7, 0.15)):
            snaps = []
            for k in range(step, len(allw) + step, step):
                snaps.append(" ".join(allw[:k]))
            if rate:
                snaps = [corrupt(rng, s, rate) for s in snaps]
            run_text.append({"snapshots": snaps, "emissions": pipe.run_on_text(snaps)})

    # ------------------------------------------------------------------ chunked --------
    st.sf.write = lambda *a, **k: None            # soundfile is an empty stand-in module here
    chunked = []
    for seq in seqs[:7]:
        allw = " ".join(by_ref[r]["text_clean"] for r in seq).split()
        for variant in range(4):
            step = rng.choice((2, 3, 4, 5))
            script = []
            for k in range(0, len(allw), step):
                t = " ".join(allw[k:k + step])
                if variant == 0:
                    script.append(t)
                elif variant == 1:
                    script.append(corrupt(rng, t, 0.1))
                else:
                    lpb = rng.choice((-0.2, -0.5, -0.9, -1.4, -2.0)) if variant == 3 else rng.choice((-0.1, -0.6))
                    tt = t if rng.random() > 0.25 else rng.choice(("", "ا", t.split()[0]))
                    script.append({"text": corrupt(rng, tt, 0.08) if tt else "", "avg_logprob": lpb})
            chunk_seconds = rng.choice((2.0, 3.0, 3.5))
            overlap = rng.choice((0.0, 0.0, 0.5))
            step_samples = max(int(chunk_seconds * 16000) - int(overlap * 16000), 1)
            tail = rng.choice((0, 5000, 9000, 20000))            # < 0.5 s tail is dropped, < 1 s is padded
            n_samples = step_samples * (len(script) - 1) + (tail if tail else int(chunk_seconds * 16000))
            st.load_audio = lambda p, n=n_samples: np.zeros(n, dtype=np.float32)
            calls = []

            def fn(path, script=script, calls=calls):
                i = len(calls)
                calls.append(i)
                return script[i] if i < len(script) else ""

            em = pipe.run_on_audio_chunked("synthetic.wav", fn, chunk_seconds=chunk_seconds, overlap_seconds=overlap)
            chunked.append({"n_samples": n_samples, "chunk_seconds": chunk_seconds, "overlap_seconds": overlap,
                            "script": script, "n_calls": len(calls), "emissions": em})

    obj = {"best_match": best, "run_on_text": run_text, "chunked": chunked}
    txt = json.dumps(obj, ensure_ascii=False, separators=(",", ":"))
    path = HERE / "tracker_cases.json.gz"
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(txt.encode("utf-8"))
    print(f"wrote {path.name}: {path.stat().st_size} bytes; best_match={len(best)} run_on_text={len(run_text)} "
          f"chunked={len(chunked)}")


if __name__ == "__main__":
    main()
