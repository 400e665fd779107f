"""Golden case for transcripts LONGER than the device's 1,024-character matching window
(StreamingPipeline.run_on_full_transcript, shared/streaming.py:58-105, has no length limit).

Run ONLY in the build container (needs /root/reference; slow -- tens of minutes: every match_verse call
slides each of the 6,236 verses over a ~1,500-character text with a pure-Python Indel ratio):

    PYTHONHASHSEED=0 python tests/golden/gen_longtx_golden.py

Writes tests/golden/longtx_cases.json: {"text", "emissions"} per case.
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path

if os.environ.get("PYTHONHASHSEED") != "0":
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))

from ref_import import load_reference  # noqa: E402


def main():
    cd = load_reference()
    db = cd._db
    from shared import streaming as st

    by_ref = {(v["surah"], v["ayah"]): v for v in db.verses}
    pipe = st.StreamingPipeline(db)
    texts = [" ".join(by_ref[(2, a)]["text_clean"] for a in (282, 283, 284, 285, 286)),
             " ".join(by_ref[(26, a)]["text_clean"] for a in range(10, 52))]
    out = []
    for text in texts:
        assert len(text) > 1024, len(text)
        em = pipe.run_on_full_transcript("x.wav", lambda p, t=text: t)
        out.append({"text": text, "chars": len(text), "emissions": em})
        print(len(text), em, flush=True)
        (HERE / "longtx_cases.json").write_text(json.dumps(out, ensure_ascii=False, separators=(",", ":")), encoding="utf-8")


if __name__ == "__main__":
    main()
