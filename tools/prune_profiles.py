#!/usr/bin/env python
"""Round-6 housekeeping of profiles/: keep, per round, the final evidence set and whatever the CURRENT documents cite; move the
rest to profiles/archive/ (git mv) and rewrite every `profiles/<name>` reference in the tracked text files to the new place.

    python tools/prune_profiles.py [--dry-run]

Kept in profiles/: README.md, everything of round 6, the final sets of rounds 3-5 (r03_m_*, r04_t_*, r04_u_*, r05_zzz_*,
r05_zzzz_*) and every file named in DESIGN.md, README.md or INTEGRATION.md.  bench.py globs profiles/r*_pmc_traffic.json /
r*_mfma_busy.json / r*_mfma_in_situ*.json: the newest of each stays by the rules above.
"""
from __future__ import annotations

import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PROF = ROOT / "profiles"
KEEP_PREFIX = ("r06_", "r05_zzz_", "r05_zzzz_", "r04_t_", "r04_u_", "r03_m_")
CURRENT_DOCS = ("DESIGN.md", "README.md", "INTEGRATION.md")


def main():
    dry = "--dry-run" in sys.argv
    cited = set()
    for d in CURRENT_DOCS:
        cited |= set(re.findall(r"profiles/([A-Za-z0-9_.\-]+)", (ROOT / d).read_text(encoding="utf-8")))
    files = [p for p in sorted(PROF.iterdir()) if p.is_file()]
    move = []
    for p in files:
        n = p.name
        if n == "README.md" or n.startswith(KEEP_PREFIX) or n in cited:
            continue
        # a cited prefix with a wildcard ("profiles/r05_a_interference_*") keeps the whole family
        if any(c.endswith("_") and n.startswith(c) for c in cited):
            continue
        move.append(n)
    print(f"{len(files)} files, {len(move)} to archive, {len(files) - len(move)} stay")
    if dry:
        return
    (PROF / "archive").mkdir(exist_ok=True)
    for n in move:
        subprocess.run(["git", "mv", str(PROF / n), str(PROF / "archive" / n)], check=True, cwd=ROOT)
    # rewrite references in tracked text files
    tracked = subprocess.run(["git", "ls-files"], capture_output=True, text=True, check=True, cwd=ROOT).stdout.split("\n")
    names = sorted(move, key=len, reverse=True)
    pat = re.compile(r"profiles/(" + "|".join(re.escape(n) for n in names) + r")(?![A-Za-z0-9_.\-])")
    changed = 0
    for t in tracked:
        # (driver-written records at the repo root -- BENCH_r*.json, GPUTEST_r*.json, VERDICT.md, ... -- are never touched)
        if not t or t.startswith("profiles/") and not t.endswith(".md") or t.startswith("tests/golden/") or (
                "/" not in t and (t.endswith(".json") or t.endswith(".jsonl") or t in ("VERDICT.md", "ADVICE.md", "SURVEY.md", "BASELINE.md"))):
            continue
        p = ROOT / t
        if not p.is_file() or p.suffix in (".gz", ".bin", ".npz", ".so", ".png"):
            continue
        try:
            s = p.read_text(encoding="utf-8")
        except (UnicodeDecodeError, OSError):
            continue
        s2 = pat.sub(lambda m: "profiles/archive/" + m.group(1), s)
        if s2 != s:
            p.write_text(s2, encoding="utf-8")
            changed += 1
    print(f"references rewritten in {changed} files")


if __name__ == "__main__":
    main()
