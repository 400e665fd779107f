"""BASELINE.json configs[1] at full size (64 clips x 10 s on one MI355X) through properties that do
not need the CPU oracle to finish a batch of that size: idempotence, permutation equivariance and
subset consistency of the whole path (all bit-exact, every stage is batch-invariant by
construction), and the verse -> log-probs -> verse round trip of the post-logits stages for 64
verses at once."""

import numpy as np
import pytest
import torch

from synth import synth_audio, synth_logits

pytestmark = pytest.mark.gpu

B, N = 64, 160000


def key(r):
    return (r["surah"], r["ayah"], r["ayah_end"], r["source"], r["score"], r["n_candidates"], r["t_frames"], r["flags"])


def test_whole_path_properties_at_batch_64():
    from offline_tarteel_amd.engine import Engine

    eng = Engine(device=0, with_model=True, seed=20260630, max_batch=B, max_samples=N)
    try:
        audio = torch.from_numpy(synth_audio(B, N)).cuda()
        lens = [N - 1280 * (b % 5) for b in range(B)]                 # a few distinct lengths, packed rows
        for b, n in enumerate(lens):
            audio[b, n:] = 0
        first = [key(r) for r in eng.predict_batch(audio, lens, want_text=False)]
        assert len(first) == B and all(k[6] == eng.frames_for(n) for k, n in zip(first, lens))
        # idempotence: nothing carries over from one call to the next
        assert [key(r) for r in eng.predict_batch(audio, lens, want_text=False)] == first
        # permutation equivariance: reversing the batch reverses the results, bit for bit
        rev = torch.flip(audio, dims=[0]).contiguous()
        assert [key(r) for r in eng.predict_batch(rev, lens[::-1], want_text=False)] == first[::-1]
        # subset consistency: an utterance's result does not depend on its batch neighbours
        sub = [5, 17, 40, 63]
        got = eng.predict_batch(audio[sub].contiguous(), [lens[i] for i in sub], want_text=False)
        assert [key(r) for r in got] == [first[i] for i in sub]
    finally:
        eng.close()


def test_verse_round_trip_at_batch_64():
    """64 seeded verses -> synthetic log-probs favouring their token ids -> greedy decode, retrieval
    and the text gate return a verse with exactly that text (identical verses exist: the refrains)."""
    from offline_tarteel_amd.engine import Engine

    eng = Engine(device=0, with_model=False, max_batch=B, max_samples=N)
    try:
        T = 126
        rng = np.random.default_rng(64)
        verses, lps = [], []
        while len(lps) < B:
            v = int(rng.integers(0, eng.tables.n_verses))
            ids = eng.tables.token_ids(v, 1).tolist()
            if not (4 <= len(ids) and 3 * len(ids) + 1 <= T):      # rep = 2 frames per token + a blank
                continue
            lg = torch.from_numpy(synth_logits(ids, T, seed=7000 + len(lps), noise=1.0, boost=8.0, rep=2))
            lps.append(torch.log_softmax(lg, -1))
            verses.append(v)
        res = eng.decode_retrieve_rerank(torch.stack(lps).cuda().contiguous(), [T] * B)
        tok = lambda v: eng.tables.token_ids(v, 1).tolist()  # noqa: E731
        for v, r in zip(verses, res):
            assert r["greedy_ids"] == tok(v)
            got = eng.tables.verse_index(r["surah"], r["ayah"])
            assert r["ayah_end"] == r["ayah"] and tok(got) == tok(v), (v, r)
            assert r["source"] == "text" and r["score"] >= 0.8
    finally:
        eng.close()


def test_int4_batch_256_is_batch_size_invariant():
    """BASELINE.json configs[2] (int4 Linear weights, 256 clips per call): the result of an utterance
    is the same bits whether it travels in a batch of 256 or of 64, and a repeated call is identical."""
    from offline_tarteel_amd.engine import Engine

    Bq = 256
    eng = Engine(device=0, with_model=True, seed=20260630, precision=1, max_batch=Bq, max_samples=N)
    try:
        base = torch.from_numpy(synth_audio(64, N)).cuda()
        audio = torch.cat([base * (1.0 - 0.01 * k) for k in range(4)], 0).contiguous()     # 256 distinct clips
        lens = [N] * Bq
        big = [key(r) for r in eng.predict_batch(audio, lens, want_text=False)]
        assert [key(r) for r in eng.predict_batch(audio, lens, want_text=False)] == big
        for k in (0, 3):
            part = eng.predict_batch(audio[64 * k: 64 * (k + 1)].contiguous(), [N] * 64, want_text=False)
            assert [key(r) for r in part] == big[64 * k: 64 * (k + 1)]
    finally:
        eng.close()


def test_configs3_rank_slices_256x10s_through_shard_plan():
    """BASELINE configs[3] (8 GPUs, global batch 2048): the slices two of the eight ranks would process --
    256 clips each, taken from a ragged 2048-clip batch by dist.shard_plan -- run at full size on this GPU;
    every clip's packed row must equal the row the clip gets in a small batch of its own (the path is
    batch-invariant), and un-permuting the slices through the all-gather's bookkeeping puts each row at its
    original position."""
    from offline_tarteel_amd import dist as qdist
    from offline_tarteel_amd.engine import Engine

    world, G, per = 8, 2048, 256
    rng = np.random.default_rng(2048)
    lengths = (160000 - 1280 * rng.integers(0, 40, size=G)).astype(np.int64)     # 6.8 .. 10 s
    order, slices = qdist.shard_plan(lengths, world)
    assert len(order) == G and all(s.stop - s.start == per for s in slices)
    base = synth_audio(64, N)

    def clip(i):   # utterance i of the global batch: one of 64 base clips, scaled, cut to its length
        a = base[i % 64] * np.float32(1.0 - 0.004 * (i // 64))
        a[lengths[i]:] = 0
        return a

    eng = Engine(device=0, with_model=True, seed=20260630, max_batch=per, max_samples=N)
    try:
        gathered = np.zeros((G, 4), np.int32)
        have = np.zeros(G, bool)
        for r in (0, 5):
            idx = order[slices[r]].tolist()
            lens = [int(lengths[i]) for i in idx]
            assert lens == sorted(lens, reverse=True)           # length-sorted inside the shard
            audio = torch.from_numpy(np.stack([clip(i) for i in idx])).cuda()
            res = eng.predict_batch(audio, lens, want_text=False)
            packed = qdist.pack_results(res)
            for k, i in enumerate(idx):
                gathered[i] = packed[k]
                have[i] = True
            # spot check: 6 clips of this slice on their own
            sub = [0, 31, 100, 177, 254, 255]
            alone = eng.predict_batch(audio[sub].contiguous(), [lens[k] for k in sub], want_text=False)
            assert qdist.pack_results(alone).tolist() == packed[sub].tolist(), r
        assert have.sum() == 2 * per
        un = qdist.unpack_results(gathered[have])
        assert all(u["surah"] >= 0 for u in un)
    finally:
        eng.close()
