"""Data-parallel glue for N GPUs of one node.

Utterances are independent, so the path shards with NO data-path collective: rank r of W takes
a contiguous slice of the (length-sorted) batch.  The only exchange is one all-gather per batch
of the packed results int32[B/W, 4] = (surah, ayah, ayah_end, float-bits(score)) -- 16 B per
utterance, latency-bound on xGMI; ``backend="nccl"`` IS RCCL on ROCm, ``"gloo"`` on CPU tests.
"""

from __future__ import annotations

import numpy as np


def shard_plan(lengths, world: int):
    """sort by length (bounds padding inside a shard), deal contiguous slices of equal size.
    Returns (order, per_rank_slices); order[k] = original index of the k-th sorted utterance.
    The batch is padded with -1 up to a multiple of world so every rank gathers the same shape."""
    lengths = np.asarray(lengths)
    order = np.argsort(-lengths, kind="stable")
    per = -(-len(order) // world)
    padded = np.full(per * world, -1, dtype=np.int64)
    padded[: len(order)] = order
    return padded, [slice(r * per, (r + 1) * per) for r in range(world)]


def pack_results(results) -> np.ndarray:
    out = np.zeros((len(results), 4), dtype=np.int32)
    for i, r in enumerate(results):
        out[i, 0], out[i, 1] = r["surah"], r["ayah"]
        out[i, 2] = r["ayah_end"] or r["ayah"]
        out[i, 3] = np.float32(r["score"]).view(np.int32)
    return out


def unpack_results(packed: np.ndarray) -> list[dict]:
    res = []
    for s, a, e, bits in packed.tolist():
        sc = float(np.int32(bits).view(np.float32))
        res.append({"surah": s, "ayah": a, "ayah_end": (e if s else None), "score": sc})
    return res


def all_gather_results(local_packed, order, n_total: int, group=None):
    """local_packed: int32 tensor [per, 4] (cuda with RCCL, cpu with gloo).  Returns the
    un-permuted int32 array [n_total, 4] on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    gathered = torch.empty((world * local_packed.shape[0], 4), dtype=torch.int32, device=local_packed.device)
    dist.all_gather_into_tensor(gathered, local_packed.contiguous(), group=group)
    g = gathered.cpu().numpy()
    out = np.zeros((n_total, 4), dtype=np.int32)
    for k, orig in enumerate(np.asarray(order).tolist()):
        if orig >= 0:
            out[orig] = g[k]
    return out
