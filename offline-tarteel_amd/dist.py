"""Data-parallel glue for N GPUs of one node.

Utterances are independent, so the path shards with NO data-path collective: the batch is sorted by length and dealt
to the ranks, every rank runs its own engine on its own GPU, and the only exchange is one all-gather per batch of the
packed results int32[per_rank, 4] = (surah, ayah, ayah_end, float32-bits(score)) -- 16 B per utterance, latency-bound on
xGMI; ``backend="nccl"`` IS RCCL on ROCm, ``"gloo"`` on CPU tests.

What ships on top of it: ``benchmark/runner.py`` (``python -m torch.distributed.run ... -m
offline_tarteel_amd.benchmark.runner``: rank 0 reads the manifest, ``shard_plan`` deals the files, every rank predicts
its share, ``all_gather_results`` restores the manifest order, rank 0 scores and writes the result files) and
``bench.py --workload strong2048`` (one ragged global batch of 2,048 clips through the same three functions).

Scores travel as float32 bits.  The plugin's scores are rounded to 4 decimals in double precision before they are
packed (reference: experiments/c2c-direct-mixed/run.py ``round(score, 4)``); a 4-decimal value in [0, 1] survives
the float32 round trip exactly once it is rounded to 4 decimals again (float32 carries 7 digits), which is what
``unpack_results(..., round_dp=4)`` does -- the runner's JSON is then bit-identical to a single-process run.
"""

from __future__ import annotations

import numpy as np


def shard_plan(lengths, world: int, deal: str = "strided"):
    """Sort by length (longest first), deal to ``world`` ranks.  Returns (order, per_rank_slices): ``order`` is the
    concatenation of the ranks' shares (rank r owns ``order[per_rank_slices[r]]``, a contiguous run), entries are
    original indices, -1 pads every share to the same size so that all ranks gather the same shape.

    deal="strided" (default): rank r takes the sorted utterances r, r + world, r + 2 world, ... -- every rank gets the
    same length distribution, so the ranks of a strong-scaling batch finish together (the slowest rank sets the batch
    time), and a rank's own share is still sorted, so consecutive engine calls see similar lengths.
    deal="contiguous": rank r takes the r-th run of the sorted list (SURVEY.md 8e's wording): least padding inside a
    rank, but rank 0 holds all the long clips -- kept for comparison (bench.py --workload strong2048 --deal contiguous)."""
    lengths = np.asarray(lengths)
    order = np.argsort(-lengths, kind="stable")
    n = len(order)
    per = -(-n // world) if n else 0
    padded = np.full(per * world, -1, dtype=np.int64)
    if deal == "contiguous":
        padded[:n] = order
    elif deal == "strided":
        for r in range(world):
            mine = order[r::world]
            padded[r * per: r * per + len(mine)] = mine
    else:
        raise ValueError(f"deal={deal!r}: expected 'strided' or 'contiguous'")
    return padded, [slice(r * per, (r + 1) * per) for r in range(world)]


def pack_results(results) -> np.ndarray:
    out = np.zeros((len(results), 4), dtype=np.int32)
    for i, r in enumerate(results):
        if not r or not r.get("surah"):
            continue
        out[i, 0], out[i, 1] = r["surah"], r["ayah"]
        out[i, 2] = r["ayah_end"] or r["ayah"]
        out[i, 3] = np.float32(r["score"]).view(np.int32)
    return out


def unpack_results(packed: np.ndarray, round_dp: int | None = None) -> list[dict]:
    res = []
    for s, a, e, bits in np.asarray(packed).tolist():
        sc = float(np.int32(bits).view(np.float32))
        if round_dp is not None:
            sc = round(sc, round_dp)
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


def all_gather_rows(local, order, n_total: int, group=None) -> np.ndarray:
    """Same un-permuting gather for a float32 side table [per, C] (the runner's per-file latency and error flag:
    harness bookkeeping, not part of the path's exchange)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    local = local.contiguous()
    gathered = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    g = gathered.cpu().numpy()
    out = np.zeros((n_total, local.shape[1]), dtype=g.dtype)
    for k, orig in enumerate(np.asarray(order).tolist()):
        if orig >= 0:
            out[orig] = g[k]
    return out


def init_process_group(backend: str | None = None):
    """torch.distributed from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*): RCCL ("nccl") when
    this rank has a GPU, gloo otherwise; QVERSE_DIST_BACKEND overrides.  Returns (rank, world, device)."""
    import os

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or os.environ.get("QVERSE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local)
        device = torch.device(f"cuda:{local}")
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=device)
    else:
        device = torch.device("cpu")
        if not dist.is_initialized():
            dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size(), device
