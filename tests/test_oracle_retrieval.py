"""Oracle retrieval stages vs fixtures from the unmodified reference (CPU)."""

import gzip
import json

import pytest

from oracle.oracle import normalize_arabic

# the reference ranks trigram candidates by iterating a hash-randomised set[str]; for this
# query many verses tie on the IDF sum, so order/membership at the top-50 cut is not a
# function of the input.  The oracle's canonical rule (ascending verse index) is checked
# for score-equivalence only.
TIE_DEPENDENT = {"garbage_40"}


@pytest.fixture(scope="module")
def cases(golden_dir):
    return json.load(gzip.open(golden_dir / "retrieval_cases.json.gz"))


def _lists(x):
    return [list(t) for t in x]


def test_retrieval_cases(oracle, cases):
    assert len(cases) >= 20
    for c in cases:
        t = c["transcript"]
        g = c["match"]
        m = oracle.match_verse(t)
        assert _lists(oracle.search(t)) == c["search100"], c["name"]
        assert _lists(oracle.pass3(t)) == c["pass3_100"], c["name"]
        assert (m["surah"], m["ayah"], m["ayah_end"], m["score"], m["raw_score"]) == (
            g["surah"], g["ayah"], g["ayah_end"], g["score"], g["raw_score"]), c["name"]
        if c["name"] in TIE_DEPENDENT:
            continue
        assert oracle.trigram_candidates(normalize_arabic(t)) == c["trigram_top50"], c["name"]
        assert _lists(m["runners_up"]) == g["runners_up"], c["name"]
        cs, cp, sc, _ = oracle.build_candidates(t)
        keys = [list(oracle.key_of(int(a), int(b))) for a, b in zip(cs, cp)]
        assert keys == c["candidates"], c["name"]
        assert sc.tolist() == c["cand_scores"], c["name"]


def test_tie_dependent_case_is_score_equivalent(oracle, cases):
    c = [x for x in cases if x["name"] == "garbage_40"][0]
    m = oracle.match_verse(c["transcript"])
    got = sorted((s for _, _, s in m["runners_up"]), reverse=True)
    want = sorted((s for _, _, s in c["match"]["runners_up"]), reverse=True)
    # same multiset of scores except at most the few entries around the tied cut
    diff = sum(1 for a, b in zip(got, want) if a != b)
    assert diff <= 8
