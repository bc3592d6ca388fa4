"""Sharding of the `yacht run` path over the GPUs of one node: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The path partitions by REFERENCE: every rank holds a contiguous range of the
references, cut so that ranks hold about the same number of hashes; the sample is replicated;
each rank runs the overlap kernel on its shard with no data-path collective, and ONE all-gather
of the per-reference uint32 counts assembles the global vector (<= 1.6 MB at 400 k references:
latency-bound, so the three count arrays travel in a single collective).

Torch is plumbing here (process group + collectives); the compute callables are RefDB methods.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def shard_plan(offsets: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous reference ranges [(begin, end)] per rank, balanced by hash count: rank r's range
    ends at the first reference boundary at or past r+1 shares of the hashes."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    total = int(offsets[-1])
    cuts = [0]
    for r in range(1, world):
        target = (total * r) // world
        j = int(np.searchsorted(offsets, np.uint64(target), side="left"))
        cuts.append(min(max(j, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def slice_csr(values: np.ndarray, offsets: np.ndarray, begin: int, end: int) -> Tuple[np.ndarray, np.ndarray]:
    """The CSR of references [begin, end) with offsets rebased to 0."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    lo, hi = int(offsets[begin]), int(offsets[end])
    return np.ascontiguousarray(values[lo:hi]), (offsets[begin:end + 1] - offsets[begin]).astype(np.uint64)


def gather_counts(local, plan: Sequence[Tuple[int, int]], group=None):
    """All-gather per-reference count rows.  `local` is a [k, n_local] integer tensor on this
    rank's device (k count arrays of the rank's shard); returns the [k, N] tensor of the whole
    database in reference order, identical on every rank.  One collective: shards are padded to
    the longest one."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(plan) == world
    lens = [e - b for b, e in plan]
    assert local.shape[-1] == lens[rank], "local counts do not match this rank's shard"
    k = local.shape[0]
    n_max = max(lens) if lens else 0
    padded = torch.zeros((k, n_max), dtype=local.dtype, device=local.device)
    padded[:, : lens[rank]] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([out[r][:, : lens[r]] for r in range(world)], dim=1)


def sharded_overlap(sample: np.ndarray, compute_local: Callable[[np.ndarray], np.ndarray],
                    plan: Sequence[Tuple[int, int]], device="cpu", group=None) -> np.ndarray:
    """overlap of `sample` with every reference of the sharded database: local kernel + one
    all-gather.  `compute_local(sample)` returns this rank's uint32 counts (RefDB.overlap)."""
    import torch

    local = np.ascontiguousarray(compute_local(sample), dtype=np.uint32)
    t = torch.from_numpy(local.view(np.int32)).to(device).unsqueeze(0)
    return gather_counts(t, plan, group=group)[0].cpu().numpy().view(np.uint32)
