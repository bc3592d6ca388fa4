"""Sharding of the `yacht run` and `yacht train` paths over the GPUs of one node: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The path partitions by REFERENCE: every rank holds a contiguous range of the
references, cut so that ranks hold about the same number of hashes; the sample is replicated;
each rank runs the overlap kernel on its shard with no data-path collective, and ONE all-gather
of the per-reference uint32 counts assembles the global vector (<= 1.6 MB at 400 k references:
latency-bound, so the three count arrays travel in a single collective).

Torch is plumbing here (process group + collectives); the compute callables are RefDB methods.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

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


# ======================================================================================================
# `yacht train` over ranks: rows of the pairwise matrix (BASELINE.json configs[3], "tiled across GPUs")
# ======================================================================================================
# Every rank holds the whole reference set (10^4-10^5 sketches fit one GPU many times over) and
# computes the ordered pairs (i, j) whose row i lies in its block: RefDB.pairwise(c, row_begin,
# row_end).  The work of a row is the number of postings its reference takes part in, so blocks are
# cut by cumulative `nshared`, not by row count.  The pair lists (a few hundred kB) are all-gathered
# once; selection (yh_train_select) then runs on the concatenated list, identically on every rank.
def pair_row_plan(nshared: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous row blocks [(begin, end)] per rank with about equal sums of nshared (+1 per row)."""
    w = np.asarray(nshared, dtype=np.int64) + 1
    n = int(w.size)
    cum = np.concatenate([[0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        j = int(np.searchsorted(cum, (int(cum[-1]) * r) // world, side="left"))
        cuts.append(min(max(j, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def sharded_pairwise(compute_rows: Callable[[int, int], Tuple[np.ndarray, np.ndarray, np.ndarray]],
                     plan: Sequence[Tuple[int, int]], device="cpu", group=None):
    """All pairs of the database, rows ascending, from per-rank row blocks.  `compute_rows(b, e)`
    returns this rank's (i, j, count) with b <= i < e, sorted by (i, j) (RefDB.pairwise).  Two
    collectives: the list lengths, then the lists padded to the longest."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(plan) == world
    b, e = plan[rank]
    pi, pj, pc = compute_rows(b, e) if e > b else (np.zeros(0, np.uint32),) * 3
    mine = torch.from_numpy(np.stack([np.asarray(x, dtype=np.int64) for x in (pi, pj, pc)], axis=0)).to(device)
    n_mine = torch.tensor([mine.shape[1]], dtype=torch.int64, device=device)
    lens = [torch.zeros_like(n_mine) for _ in range(world)]
    dist.all_gather(lens, n_mine, group=group)
    lens = [int(x.item()) for x in lens]
    n_max = max(lens) if lens else 0
    padded = torch.zeros((3, max(n_max, 1)), dtype=torch.int64, device=device)
    padded[:, : mine.shape[1]] = mine
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    allp = torch.cat([out[r][:, : lens[r]] for r in range(world)], dim=1).cpu().numpy()
    return allp[0].astype(np.uint32), allp[1].astype(np.uint32), allp[2].astype(np.uint32)


_SIGN = -(2 ** 63)  # int64 bit pattern 0x8000...: x ^ _SIGN turns unsigned order into signed order


# ---- `yacht train` by HASH RANGE: the index build divides too --------------------------------------------------------
# With row blocks every rank still uploads, sorts and indexes the WHOLE reference set (13 of configs[3]'s 17 ms) and only
# the 3.6 ms pairwise pass shrinks.  By hash range, rank g holds every reference cut down to [lo_g, hi_g): it uploads,
# sorts and indexes 1/G of the (hash, reference) pairs, and its pairwise pass gives every pair's intersection count
# RESTRICTED to the range -- |R_i n R_j| is the sum of those over the ranges.  Only pairs that share a hash at all
# exist (a few per reference), so the partial lists are small: one all-gather of (i, j, partial count), a sum per pair,
# then the reference's filter `!(1.0 * count / |R_i| < C)` (src/cpp/main.cpp:297-303) on the TOTAL count with the
# WHOLE sketch's size.  The three index statistics (main.cpp:242-244) are sums over the ranges too.
def slice_csr_to_hash_range(values: np.ndarray, offsets: np.ndarray, lo: int, hi: int) -> Tuple[np.ndarray, np.ndarray]:
    """Host form of slice_to_hash_range: every reference's hashes in [lo, hi) (hi = 2**64: no upper bound)."""
    values = np.asarray(values, dtype=np.uint64)
    offsets = np.asarray(offsets, dtype=np.uint64)
    keep = values >= np.uint64(lo)
    if hi < 2 ** 64:
        keep &= values < np.uint64(hi)
    csum = np.concatenate([[0], np.cumsum(keep, dtype=np.uint64)])
    return np.ascontiguousarray(values[keep]), csum[offsets.astype(np.int64)].astype(np.uint64)


def merge_partial_pairs(n_refs: int, parts: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray]], sizes: np.ndarray,
                        c_thresh: float):
    """Sum the per-range (i, j, count) lists by pair and keep a pair iff !(1.0 * count / |R_i| < c_thresh): the pairs
    RefDB.pairwise(c_thresh) gives on the whole sketches, sorted by (i, j)."""
    if not parts or not sum(int(p[0].size) for p in parts):
        z = np.zeros(0, dtype=np.uint32)
        return z, z.copy(), z.copy()
    key = np.concatenate([p[0].astype(np.int64) * n_refs + p[1].astype(np.int64) for p in parts])
    cnt = np.concatenate([p[2].astype(np.int64) for p in parts])
    uk, inv = np.unique(key, return_inverse=True)
    tot = np.bincount(inv, weights=cnt.astype(np.float64), minlength=uk.size).astype(np.int64)  # (exact: counts << 2^53)
    pi, pj = (uk // n_refs), (uk % n_refs)
    keep = ~((1.0 * tot / np.asarray(sizes, dtype=np.float64)[pi]) < c_thresh)
    return pi[keep].astype(np.uint32), pj[keep].astype(np.uint32), tot[keep].astype(np.uint32)


def merge_partial_pairs_device(n_refs: int, allp, sizes: np.ndarray, c_thresh: float):
    """merge_partial_pairs on the device: allp = [3, total] int64 tensor (i, j, partial count) of all ranks' lists."""
    import torch

    if allp.shape[1] == 0:
        z = np.zeros(0, dtype=np.uint32)
        return z, z.copy(), z.copy()
    uk, inv = torch.unique(allp[0] * n_refs + allp[1], return_inverse=True)
    tot = torch.zeros(uk.numel(), dtype=torch.int64, device=allp.device).scatter_add_(0, inv, allp[2])
    pi_t = torch.div(uk, n_refs, rounding_mode="floor")
    sz = torch.from_numpy(np.asarray(sizes, dtype=np.float64)).to(allp.device)
    keep = ~((1.0 * tot.to(torch.float64) / sz[pi_t]) < c_thresh)
    res = torch.stack([pi_t[keep], (uk - pi_t * n_refs)[keep], tot[keep]]).cpu().numpy()
    return res[0].astype(np.uint32), res[1].astype(np.uint32), res[2].astype(np.uint32)


def hash_range_pairwise(compute_range: Callable[[], Tuple[np.ndarray, np.ndarray, np.ndarray, Tuple[int, int, int]]],
                        n_refs: int, sizes: np.ndarray, c_thresh: float, device="cpu", group=None):
    """All pairs of the database from per-rank HASH RANGES.  `compute_range()` returns this rank's (i, j, partial count)
    of every pair sharing a hash in its range (RefDB.pairwise(0.0) on the handle over the range slices) and its
    (distinct, singletons, index) statistics.  Two collectives: the list lengths (+ the statistics), then the lists.
    Returns (pair_i, pair_j, count, (distinct, singletons, index)) of the WHOLE database, identical on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    pi, pj, pc, stats = compute_range()
    if world == 1:
        gi, gj, gc = merge_partial_pairs(n_refs, [(pi, pj, pc)], sizes, c_thresh)
        return gi, gj, gc, tuple(int(x) for x in stats)
    mine = torch.from_numpy(np.stack([np.asarray(x, dtype=np.int64) for x in (pi, pj, pc)], axis=0)).to(device)
    head = torch.tensor([mine.shape[1], *[int(x) for x in stats]], dtype=torch.int64, device=device)
    heads = [torch.zeros_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    lens = [int(h[0].item()) for h in heads]
    tot_stats = tuple(int(sum(int(h[k].item()) for h in heads)) for k in (1, 2, 3))
    padded = torch.zeros((3, max(max(lens), 1)), dtype=torch.int64, device=device)
    padded[:, : mine.shape[1]] = mine
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    if out[0].is_cuda:  # sum per pair and threshold on the device (a few 10^5 entries: a sort there, not on the host)
        gi, gj, gc = merge_partial_pairs_device(n_refs, torch.cat([out[r][:, : lens[r]] for r in range(world)], dim=1), sizes, c_thresh)
        return gi, gj, gc, tot_stats
    parts = [tuple(out[r][k, : lens[r]].cpu().numpy() for k in range(3)) for r in range(world)]
    gi, gj, gc = merge_partial_pairs(n_refs, parts, sizes, c_thresh)
    return gi, gj, gc, tot_stats


# ======================================================================================================
# The `yacht run` step over ranks with ONE small exchange: ghosts (include/yacht_hip.h, "references
# spread over several GPUs, the `yacht run` subset")
# ======================================================================================================
# overlap_j needs only rank-local data, but "hash h of R_j is held by no other overlapping reference" is a
# global statement: the other holder may live on another rank.  `yacht run` only ever asks for the subset
# "overlap > 0" (hypothesis_recovery_src.py:361-378), and for that subset a rank can finish its own
# references alone once its index knows every holder of every hash it holds -- including holders on
# other ranks -- and one bit per foreign holder: does it overlap the sample at all?  So, once per
# database, every rank sends its (hash, reference) pairs to the owner of the hash's range (all_to_all), and the
# owners send back, for every hash with holders on
# more than one rank, the foreign postings to each holder's rank; a rank appends those foreign
# references as GHOSTS (extra references holding only such hashes) behind its own, and the library
# builds the usual handle over the lot.  Per sample: local lookup + reduce -> all-gather of the subset
# bits (N_total / 8 bytes; latency-bound over xGMI) -> ghosts' bits patched, posting-list part added.
# The per-reference counts leave in one final gather, which the caller may overlap with the next sample.
def batch_planes(n_samples: int) -> int:
    """Planes of 64 samples the subset words of a batch take (include/yacht_hip.h: YH_BATCH_PLANES; up to 256 samples = 4)."""
    return (int(n_samples) + 63) // 64


def _planes_of_words(words_t, n_refs: int) -> int:
    """... of a word tensor [planes * N] (or [1, planes * N])."""
    return max(1, int(words_t.numel()) // max(int(n_refs), 1))


def _is_gloo(group) -> bool:
    import torch.distributed as dist

    return dist.get_backend(group) == "gloo"


def _stage(t, group):
    """gloo moves CPU tensors: device tensors are staged through the host there (CPU tests, and the
    two-processes-on-one-GPU test); RCCL takes the device tensor as it is."""
    return t.cpu() if (t.is_cuda and _is_gloo(group)) else t


def all_gather_into(out_t, in_t, group=None):
    """dist.all_gather_into_tensor on the current stream (staged through the host for gloo)."""
    import torch.distributed as dist

    if in_t.is_cuda and _is_gloo(group):
        o = out_t.cpu()
        dist.all_gather_into_tensor(o, in_t.cpu().contiguous(), group=group)
        out_t.copy_(o)
        return
    dist.all_gather_into_tensor(out_t, in_t.contiguous(), group=group)


def all_to_all_v(in_t, send_counts, group=None):
    """Variable all-to-all of a 1-D tensor cut at `send_counts`: (received tensor, recv_counts)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    src = _stage(in_t, group).contiguous()
    send_t = torch.tensor(list(send_counts), dtype=torch.int64, device=src.device)
    recv_t = torch.zeros(world, dtype=torch.int64, device=src.device)
    dist.all_to_all_single(recv_t, send_t, group=group)
    recv = [int(x) for x in recv_t.tolist()]
    out = torch.empty(sum(recv), dtype=in_t.dtype, device=src.device)
    dist.all_to_all_single(out, src, recv, list(send_counts), group=group)
    return out.to(in_t.device), recv


class HipLocalBackend:
    """The compute side of ShardedRefDB on the HIP engine: one RefDB over local references + ghosts.
    Everything is queued on torch's CURRENT stream (the handle is pointed at it), so the collectives
    torch issues on that stream are ordered with the kernels without host synchronisation."""

    def __init__(self, device_index: int):
        self.device_index = device_index

    def make_local_db(self, values_t, offsets_t, ghost_begin: int, ghost_src_t):
        import torch

        from .engine import RefDB

        n = offsets_t.numel() - 1
        torch.cuda.current_stream().synchronize()  # build-time: the CSR tensors are complete
        db = RefDB.from_device(values_t.data_ptr(), offsets_t.data_ptr(), n, device=self.device_index)
        n_ghost = int(ghost_src_t.numel())
        if n_ghost:
            db.set_ghosts(ghost_begin, n_ghost, ghost_src_t.data_ptr())
        backend = self

        class _Local:
            handle = db

            def bind_stream(self):
                db.set_stream(torch.cuda.current_stream().cuda_stream)

            def run_local(self, sample_t, counts_t, bits_t, ctx=0):
                db.run_local_device(sample_t.data_ptr(), sample_t.numel(), counts_t[0].data_ptr(), counts_t[1].data_ptr(),
                                    counts_t[2].data_ptr(), bits_t.data_ptr(), ctx)

            def run_finish(self, global_bits_t, counts_t, ctx=0):
                db.run_finish_device(global_bits_t.data_ptr(), counts_t[1].data_ptr(), ctx)

            def close(self):
                db.close()

        loc = _Local()
        loc.bind_stream()
        return loc


class ShardedRefDB:
    """`yacht run` counts for references spread over the ranks of `group`, subset = overlap > 0.

    values_t / offsets_t: this rank's references (CSR, offsets rebased to 0) as int64 tensors holding
    the uint64 bit patterns, on this rank's device.  Results per rank: a [3, n_pad] int32 tensor whose
    first n_local columns are overlap / n_exclusive / n_matches of ITS references; gather() assembles
    the [3, N_total] table in reference order on every rank."""

    def __init__(self, values_t, offsets_t, backend, group=None, block: int = 1):
        """block: samples whose subset bits travel in ONE all-gather (begin / exchange / end below).  A torch
        collective costs the host tens of microseconds and the device two cross-queue hand-overs, more than a
        whole step's kernels are worth; with block = 8 a sample pays an eighth of that.  Two blocks are in flight."""
        import torch
        import torch.distributed as dist

        assert 1 <= block <= 8, "two blocks of `block` samples use 2 * block of the library's 16 step contexts"
        self.block = GB = int(block)
        self.group = group
        self.world = world = dist.get_world_size(group)
        self.rank = rank = dist.get_rank(group)
        dev = values_t.device
        self.dev = dev
        n_local = int(offsets_t.numel() - 1)
        self.n_local = n_local
        sizes_local = (offsets_t[1:] - offsets_t[:-1])

        lens_t = torch.zeros(world, dtype=torch.int64, device="cpu" if _is_gloo(group) else dev)
        dist.all_gather_into_tensor(lens_t, torch.tensor([n_local], dtype=torch.int64, device=lens_t.device), group=group)
        lens = [int(x) for x in lens_t.tolist()]
        starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        self.plan = [(int(starts[r]), int(starts[r + 1])) for r in range(world)]
        self.n_total = int(starts[-1])
        # bit space of the exchange: rank r's reference i is bit r * BITS + i
        self.words = 2 * ((max(lens + [1]) + 63) // 64)
        BITS = self.words * 32
        n_pad = ((n_local + 63) // 64) * 64

        ghost_vals = torch.zeros(0, dtype=torch.int64, device=dev)
        ghost_sizes = torch.zeros(0, dtype=torch.int64, device=dev)
        ghost_src = torch.zeros(0, dtype=torch.int32, device=dev)
        # (YH_FORCE_EXCHANGE=1: run every collective of the build and of the step with ONE rank too -- how the
        # RCCL call shapes and dtypes are exercised on a 1-GPU box)
        self.has_exchange = world > 1 or os.environ.get("YH_FORCE_EXCHANGE") == "1"
        if self.has_exchange:
            # (hash, global bit id) pairs in hash order, cut at the owners' range bounds
            key = values_t ^ _SIGN  # signed order == unsigned hash order
            top = key.max().reshape(1).clone() if values_t.numel() else torch.tensor([_SIGN], dtype=torch.int64, device=dev)
            top_c = _stage(top, group)
            dist.all_reduce(top_c, op=dist.ReduceOp.MAX, group=group)
            max_hash = int(top_c.item() ^ _SIGN) & (2 ** 64 - 1)
            gid = torch.repeat_interleave(torch.arange(n_local, device=dev, dtype=torch.int64) + rank * BITS, sizes_local)
            key, perm = torch.sort(key, stable=True)
            gid = gid[perm]
            del perm
            bounds = [(((max_hash + 1) * d) // world) for d in range(1, world)]
            bkeys = torch.tensor([(b - 2 ** 63) for b in bounds], dtype=torch.int64, device=dev)
            cuts = [0] + ([int(c) for c in torch.searchsorted(key, bkeys).tolist()] if world > 1 else []) + [int(key.numel())]
            send = [cuts[d + 1] - cuts[d] for d in range(world)]
            h_in, _ = all_to_all_v(key, send, group)       # (still in signed-order form)
            g_in, _ = all_to_all_v(gid, send, group)
            del key, gid
            # owner side: runs of equal hashes whose holders live on more than one rank
            h_in, perm = torch.sort(h_in, stable=True)
            g_in = g_in[perm]
            del perm
            n_in = int(h_in.numel())
            if n_in:
                head = torch.ones(n_in, dtype=torch.bool, device=dev)
                head[1:] = h_in[1:] != h_in[:-1]
                run = torch.cumsum(head.to(torch.int64), 0) - 1
                n_runs = int(run[-1].item()) + 1
                r_of = torch.div(g_in, BITS, rounding_mode="floor")
                present = torch.zeros((n_runs, world), dtype=torch.bool, device=dev)
                present[run, r_of] = True
                multi = present.sum(dim=1) > 1
            back_h, back_g, back_n = [], [], []
            for d in range(world):
                if n_in:
                    sel = multi[run] & present[run, d] & (r_of != d)
                    back_h.append(h_in[sel])
                    back_g.append(g_in[sel])
                else:
                    back_h.append(h_in[:0])
                    back_g.append(g_in[:0])
                back_n.append(int(back_h[-1].numel()))
            fh, _ = all_to_all_v(torch.cat(back_h), back_n, group)
            fg, _ = all_to_all_v(torch.cat(back_g), back_n, group)
            del h_in, g_in, back_h, back_g
            if fh.numel():
                # ghosts: foreign references ordered by global id, their hashes ascending
                fh, perm = torch.sort(fh, stable=True)
                fg = fg[perm]
                fg, perm = torch.sort(fg, stable=True)
                fh = fh[perm]
                uniq, cnt = torch.unique_consecutive(fg, return_counts=True)
                ghost_vals = fh ^ _SIGN
                ghost_sizes = cnt
                ghost_src = uniq  # (global ids rank * BITS + i for now: the gathered layout is fixed below)
        self.n_ghost = int(ghost_src.numel())
        pad = torch.zeros(n_pad - n_local, dtype=torch.int64, device=dev)
        all_sizes = torch.cat([sizes_local, pad, ghost_sizes])
        offs = torch.zeros(all_sizes.numel() + 1, dtype=torch.int64, device=dev)
        offs[1:] = torch.cumsum(all_sizes, 0)
        vals = torch.cat([values_t, ghost_vals]).contiguous()
        self.n_rows = int(all_sizes.numel())
        # One row of subset bits per sample: W words, the same on every rank (the reducer writes whole 256-reference
        # blocks of this rank's rows, ghosts included).  A block leaves as [GB, W] and arrives as [world, GB, W]:
        # sample g's bit of reference i of rank r is bit (r * GB * W * 32 + i) counted from ITS row g of rank 0.
        w_t = torch.tensor([max(self.words, 2 * ((self.n_rows + 255) // 256) * 4 + 2)], dtype=torch.int64, device=dev)
        w_c = _stage(w_t, group)
        if self.has_exchange:
            dist.all_reduce(w_c, op=dist.ReduceOp.MAX, group=group)
        self.W = W = int(w_c.item())
        if self.n_ghost:
            r_of = torch.div(ghost_src, BITS, rounding_mode="floor")
            ghost_src = (r_of * (GB * W * 32) + (ghost_src - r_of * BITS)).to(torch.int32)
        else:
            ghost_src = ghost_src.to(torch.int32)
        self.local = backend.make_local_db(vals, offs.contiguous(), n_pad, ghost_src.contiguous())
        self._keep = (vals, offs, ghost_src)
        # two blocks of step contexts: the lookups of block b+1 are queued while the bit exchange of block b is in flight
        self.bits_local = [torch.zeros((GB, W), dtype=torch.int32, device=dev) for _ in range(2)]
        self.bits_global = [torch.zeros((world, GB, W), dtype=torch.int32, device=dev) for _ in range(2)]
        self._global_rows = [[self.bits_global[s].view(-1)[g * W:] for g in range(GB)] for s in range(2)]
        self._pending = [None, None]

    def new_counts(self):
        import torch

        return torch.zeros((3, self.n_rows), dtype=torch.int32, device=self.dev)

    def begin(self, sample_t, counts_t, slot: int = 0, g: int = 0):
        """Rank-local half of sample g of block `slot` (0 or 1): lookup + reduce; its subset bits land in row g."""
        self.local.run_local(sample_t, counts_t, self.bits_local[slot][g], slot * self.block + g)

    def exchange(self, slot: int = 0):
        """All-gather of the block's subset bits, STARTED (RCCL: asynchronously, ordered behind the kernels)."""
        import torch.distributed as dist

        if not self.has_exchange:
            return
        mine = self.bits_local[slot]
        if mine.is_cuda and not _is_gloo(self.group):
            self._pending[slot] = dist.all_gather_into_tensor(self.bits_global[slot], mine, group=self.group, async_op=True)
        else:
            all_gather_into(self.bits_global[slot].view(-1), mine.view(-1), group=self.group)

    def end(self, counts_t, slot: int = 0, g: int = 0):
        """Second half of sample g of block `slot`: wait (on the stream) for the block's bits, ghosts take their owners'
        bits, posting-list pass."""
        if self._pending[slot] is not None:
            self._pending[slot].wait()
            self._pending[slot] = None
        self.local.run_finish(self._global_rows[slot][g] if self.has_exchange else self.bits_local[slot][g], counts_t,
                              slot * self.block + g)
        return counts_t

    def run_begin(self, sample_t, counts_t, slot: int = 0):
        """One sample per exchange (block = 1): first half + the start of its bit exchange.  run_end(slot) finishes."""
        assert self.block == 1, "a ShardedRefDB made with block > 1 is driven with begin / exchange / end"
        self.begin(sample_t, counts_t, slot, 0)
        self.exchange(slot)

    def run_end(self, counts_t, slot: int = 0):
        return self.end(counts_t, slot, 0)

    def run(self, sample_t, counts_t=None):
        """One sample: this rank's [3, n_rows] counts (columns [0, n_local) are its references).
        Stream-ordered on the current stream; no host synchronisation with RCCL."""
        if counts_t is None:
            counts_t = self.new_counts()
        self.begin(sample_t, counts_t, 0, 0)
        self.exchange(0)  # (block > 1: the other rows of the block travel along, unused)
        return self.end(counts_t, 0, 0)

    def gather(self, counts_t):
        """[3, N_total] on every rank from the per-rank rows (one collective, shards padded)."""
        mine = _stage(counts_t[:, : self.n_local], self.group).contiguous()
        return gather_counts(mine, self.plan, group=self.group).to(self.dev)

    def close(self):
        self.local.close()


# ======================================================================================================
# The `yacht run` step over ranks by HASH RANGE (option B; include/yacht_hip.h)
# ======================================================================================================
# Sharding by reference (ShardedRefDB above) divides the DATABASE but not the work of the default lookup: that one
# costs per SAMPLE hash, and every rank looks up the whole sample.  Here rank g owns the hashes in [lo_g, hi_g):
# its handle holds all N references, each cut down to that range (a contiguous piece of every sorted sketch), and
# it looks up only the sample's hashes in the range (a contiguous slice of the sorted sample).  Tables and lookups
# shrink by the number of ranks.  All holders of a hash sit on its rank, so multiplicity and exclusivity are
# rank-local -- no ghosts -- and every count is a sum over the ranks once the subset is the global one:
#   begin     yh_run_local_range_device: this rank's share of overlap and n_match, its local subset bits
#   exchange  ONE all-gather of the bits of a block of samples (N/8 bytes per sample and rank)
#   end       yh_run_finish_range_device: subset = OR of the ranks' bits; this rank's share of n_exclusive
#   reduce    ONE sum of the block's [block, 3, N] count rows over the ranks (to rank 0, where results are consumed)
# ShardedRefDB stays the CAPACITY mode (a database that does not fit one GPU's HBM N times over in pieces of all
# references is still cut by reference there); this is the THROUGHPUT mode.
def hash_range_bounds(max_hash: int, world: int) -> List[int]:
    """world + 1 cut points of [0, max_hash]: equal-width ranges (FracMinHash hashes are uniform below max_hash);
    the last bound is 2**64 (exclusive upper end of the last range)."""
    return [((int(max_hash) + 1) * r) // world for r in range(world)] + [2 ** 64]


def _u64_key(x: int) -> int:
    """The int64 whose signed order equals the unsigned order of x (x ^ 2**63 as a signed number)."""
    return (int(x) ^ (2 ** 63)) - (2 ** 64 if (int(x) ^ (2 ** 63)) >= 2 ** 63 else 0)


def slice_to_hash_range(values_t, offsets_t, lo: int, hi: int):
    """The CSR of the same references holding only their hashes in [lo, hi) (hi = 2**64: no upper bound).  int64
    tensors holding uint64 bit patterns, any device; every reference's hashes ascending (unsigned)."""
    import torch

    key = values_t ^ _SIGN  # signed order == unsigned order
    keep = key >= _u64_key(lo)
    if hi < 2 ** 64:
        keep &= key < _u64_key(hi)
    csum = torch.zeros(values_t.numel() + 1, dtype=torch.int64, device=values_t.device)
    csum[1:] = torch.cumsum(keep.to(torch.int64), 0)
    return values_t[keep].contiguous(), csum[offsets_t].contiguous()


def sample_slice(sample_t, lo: int, hi: int) -> Tuple[int, int]:
    """[a, b): the positions of the sorted sample's hashes in [lo, hi)."""
    import torch

    key = sample_t ^ _SIGN
    b = [_u64_key(lo)] + ([_u64_key(hi)] if hi < 2 ** 64 else [])
    cut = torch.searchsorted(key, torch.tensor(b, dtype=torch.int64, device=sample_t.device)).tolist()
    return int(cut[0]), (int(cut[1]) if hi < 2 ** 64 else int(sample_t.numel()))


class HipRangeBackend:
    """The compute side of HashRangeRefDB on the HIP engine (everything on torch's CURRENT stream)."""

    def __init__(self, device_index: int):
        self.device_index = device_index

    def make_range_db(self, values_t, offsets_t):
        import torch

        from .engine import RefDB

        n = offsets_t.numel() - 1
        torch.cuda.current_stream().synchronize()  # build-time: the CSR tensors are complete
        db = RefDB.from_device(values_t.data_ptr(), offsets_t.data_ptr(), n, device=self.device_index)
        empty = int(values_t.numel()) == 0

        class _Range:
            handle = db

            def bind_stream(self):
                db.set_stream(torch.cuda.current_stream().cuda_stream)

            def set_finish_stream(self, stream):
                """torch.cuda.Stream (or None) for the second halves of the batched calls: what runs under
                `with torch.cuda.stream(stream)` on the torch side runs there inside the library too."""
                db.set_batch_finish_stream(None if stream is None else stream.cuda_stream)

            def run_local(self, sample_t, a, b, counts_t, bits_t, ctx=0):
                if empty:  # a range that holds no reference hash: nothing can match
                    counts_t.zero_()
                    bits_t.zero_()
                    return
                db.run_local_range_device(sample_t.data_ptr() + 8 * a, b - a, counts_t[0].data_ptr(), counts_t[2].data_ptr(),
                                          bits_t.data_ptr(), ctx)

            def run_finish(self, gathered_t, n_ranks, stride_words, counts_t, ctx=0):
                if empty:
                    return
                db.run_finish_range_device(gathered_t.data_ptr(), n_ranks, stride_words, counts_t[1].data_ptr(), ctx)

            def batch_local(self, cat_t, soff_t, n_samples, total, ov_t, words_t, slot=0):
                if empty:  # (a range without a single reference hash still takes part: its words are zero, its shares too)
                    ov_t.zero_()
                    words_t.zero_()
                    return
                db.run_batch_local_range_device(cat_t.data_ptr(), soff_t.data_ptr(), n_samples, total, ov_t.data_ptr(), words_t.data_ptr(), slot)

            def words_pack(self, words_t, packed_t, cap):
                """packed_t [words_packed_len(cap)] int64 = count, this rank's non-zero subset words, their ids (words_t: [planes * N])."""
                db.run_batch_words_pack_device(words_t.data_ptr(), packed_t.data_ptr(), int(cap), n_planes=_planes_of_words(words_t, db.n_refs))

            def words_unpack(self, gathered_t, n_ranks, cap, words_out_t, overflow_t):
                """words_out_t [planes * N] int64 = OR of the ranks' packed words; overflow_t [1] int32 = some rank had more than cap."""
                db.run_batch_words_unpack_device(gathered_t.data_ptr(), int(n_ranks), int(cap), words_out_t.data_ptr(), overflow_t.data_ptr(),
                                                 n_planes=_planes_of_words(words_out_t, db.n_refs))

            def batch_finish(self, n_samples, gathered_t, n_ranks, ov_t, e_t, m_t, slot=0):
                if empty:
                    e_t.zero_()
                    m_t.zero_()
                    self._empty_gathered = getattr(self, "_empty_gathered", {})
                    self._empty_gathered[slot] = (gathered_t, n_ranks, n_samples)  # (per batch slot: ADVICE r04)
                    return
                db.run_batch_finish_range_device(n_samples, gathered_t.data_ptr(), n_ranks, ov_t.data_ptr(), e_t.data_ptr(), m_t.data_ptr(), slot)

            def rows_pack(self, counts_t, vals_t, nrows_t, slot=0):
                """vals_t [cap, 3] int32 = this rank's shares of every entry of the slot's batch; nrows_t [1] their number."""
                if empty:  # every share is zero; the entries are those of the global subset all the same
                    g, nr, ns = self._empty_gathered[slot]  # (rare: tests; through the host)
                    w = np.bitwise_or.reduce(g.reshape(g.shape[0], -1)[:nr].cpu().numpy().view(np.uint64), axis=0)
                    n = int(np.unpackbits(w.view(np.uint8)).sum())
                    nrows_t.copy_(torch.tensor([n], dtype=torch.int32))
                    vals_t.zero_()
                    return
                db.run_batch_rows_pack_device(counts_t[0].data_ptr(), counts_t[1].data_ptr(), counts_t[2].data_ptr(), vals_t.data_ptr(),
                                              int(vals_t.shape[0]), nrows_t.data_ptr(), slot)

            def rows_unpack(self, vals_t, rows_t, nrows_t, slot=0):
                """rows_t [cap, 5] int32 = (sample, ref, overlap, n_excl, n_match) from (summed) values."""
                assert not empty, "the consumer of the rows needs a rank whose range holds reference hashes"
                db.run_batch_rows_unpack_device(vals_t.data_ptr(), int(vals_t.shape[0]), rows_t.data_ptr(), nrows_t.data_ptr(), slot)

            def close(self):
                db.close()

        r = _Range()
        r.bind_stream()
        return r


class HashRangeRefDB:
    """`yacht run` counts with the HASH SPACE spread over the ranks of `group`, subset = overlap > 0.

    values_t / offsets_t: ALL references (CSR, int64 tensors holding uint64 bit patterns, this rank's device) already
    cut down to this rank's range [bounds[rank], bounds[rank + 1]) -- slice_to_hash_range does that -- so offsets_t has
    N_total + 1 entries on every rank.  Per rank a step leaves its SHARE of the three count rows ([3, N_total] int32);
    reduce() sums the shares."""

    def __init__(self, values_t, offsets_t, bounds: Sequence[int], backend, group=None, block: int = 1):
        import torch
        import torch.distributed as dist

        assert 1 <= block <= 8, "two blocks of `block` samples use 2 * block of the library's 16 step contexts"
        self.block = GB = int(block)
        self.group = group
        self.world = world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert len(bounds) == world + 1
        self.lo, self.hi = int(bounds[rank]), int(bounds[rank + 1])
        self.dev = dev = values_t.device
        self.n_total = self.n_rows = self.n_local = int(offsets_t.numel() - 1)
        self.n_ghost = 0
        self.has_exchange = world > 1 or os.environ.get("YH_FORCE_EXCHANGE") == "1"
        # one row of subset bits per sample: whole 256-reference blocks (what the reducer writes), the same on every rank
        self.W = W = ((self.n_total + 255) // 256) * 8
        self.local = backend.make_range_db(values_t, offsets_t)
        self._keep = (values_t, offsets_t)
        self.bits_local = [torch.zeros((GB, W), dtype=torch.int32, device=dev) for _ in range(2)]
        self.bits_global = [torch.zeros((world, GB, W), dtype=torch.int32, device=dev) for _ in range(2)]
        self._pending = [None, None]

    def new_counts(self):
        import torch

        return torch.zeros((3, self.n_total), dtype=torch.int32, device=self.dev)

    def slice_of(self, sample_t) -> Tuple[int, int]:
        """[a, b): the positions of the sorted sample's hashes in this rank's range -- one small device search and a host
        sync, computed on EVERY call (nothing is remembered about a tensor: an address says nothing about the hashes
        behind it once a staging buffer is refilled or the allocator hands the address out again).  A caller whose
        samples stay resident computes the spans once and passes them to begin() / pack_batch()."""
        return sample_slice(sample_t, self.lo, self.hi) if sample_t.numel() else (0, 0)

    def begin(self, sample_t, counts_t, slot: int = 0, g: int = 0, span: Optional[Tuple[int, int]] = None):
        """Rank-local half of sample g of block `slot`: lookup + reduce of the sample's hashes in this rank's range
        (`span` = slice_of(sample_t) when the caller already has it)."""
        a, b = self.slice_of(sample_t) if span is None else span
        self.local.run_local(sample_t, a, b, counts_t, self.bits_local[slot][g], slot * self.block + g)

    def exchange(self, slot: int = 0):
        """All-gather of the block's local subset bits, STARTED (RCCL: asynchronously, ordered behind the kernels)."""
        import torch.distributed as dist

        if not self.has_exchange:
            return
        mine = self.bits_local[slot]
        if mine.is_cuda and not _is_gloo(self.group):
            self._pending[slot] = dist.all_gather_into_tensor(self.bits_global[slot], mine, group=self.group, async_op=True)
        else:
            all_gather_into(self.bits_global[slot].view(-1), mine.view(-1), group=self.group)

    def end(self, counts_t, slot: int = 0, g: int = 0):
        """Second half of sample g of block `slot`: global subset = OR of the ranks' bits, exclusive pass."""
        if self._pending[slot] is not None:
            self._pending[slot].wait()
            self._pending[slot] = None
        if self.has_exchange:  # rank r's row of sample g starts (r * block + g) * W words in
            rows = self.bits_global[slot].view(-1)[g * self.W:]
            self.local.run_finish(rows, self.world, self.block * self.W, counts_t, slot * self.block + g)
        else:
            self.local.run_finish(self.bits_local[slot][g], 1, self.W, counts_t, slot * self.block + g)
        return counts_t

    def run(self, sample_t, counts_t=None):
        """One sample: this rank's share of the [3, N_total] counts.  Stream-ordered; no host synchronisation with RCCL."""
        if counts_t is None:
            counts_t = self.new_counts()
        self.begin(sample_t, counts_t, 0, 0)
        self.exchange(0)  # (block > 1: the other rows of the block travel along, unused)
        return self.end(counts_t, 0, 0)

    # ---- many samples per call: the throughput form (include/yacht_hip.h, yh_run_batch_*_range_device) ----
    def pack_batch(self, samples, spans=None):
        """This rank's slices of up to 256 samples, concatenated: (hashes, offsets[len + 1], total) -- made once for
        samples that stay resident (`spans`: their slice_of() results, when the caller has them)."""
        import torch

        parts, lens = [], []
        for k_, s_ in enumerate(samples):
            a, b = self.slice_of(s_) if spans is None else spans[k_]
            parts.append(s_[a:b])
            lens.append(b - a)
        cat = torch.cat(parts).contiguous() if sum(lens) else torch.zeros(1, dtype=torch.int64, device=self.dev)[:0]
        soff = torch.zeros(len(samples) + 1, dtype=torch.int64, device=self.dev)
        soff[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64, device=self.dev), 0)
        return cat, soff, int(sum(lens))

    def batch_begin(self, batch, counts_t, words_t, slot: int = 0):
        """First half for a whole batch: counts_t [3, B, N] (row 0 = this rank's share of the overlaps), words_t
        [batch_planes(B) * N] int64 = this rank's subset words (plane w, bit s: sample 64 w + s overlaps the reference in this range).  `slot` (0 .. YH_BATCH_SLOTS - 1)
        names the batch slot of the library the two halves share: the words of one block travel while the next block's first
        half runs in another slot."""
        cat, soff, total = batch
        self.local.batch_local(cat, soff, int(soff.numel()) - 1, total, counts_t[0], words_t, slot)

    def batch_exchange(self, words_t, gathered_t, async_op: bool = False):
        """All-gather of the ranks' subset words into gathered_t [world, planes * N]."""
        import torch.distributed as dist

        if not self.has_exchange:
            gathered_t[0].copy_(words_t)
            return None
        if words_t.is_cuda and not _is_gloo(self.group):
            return dist.all_gather_into_tensor(gathered_t, words_t, group=self.group, async_op=async_op)
        all_gather_into(gathered_t.view(-1), words_t.view(-1), group=self.group)
        return None

    def batch_end(self, n_samples, gathered_t, counts_t, slot: int = 0, n_ranks: Optional[int] = None):
        """Second half: subset = OR of the n_ranks word rows of gathered_t (default: one per rank; 1 = a row that is already
        the OR, what batch_words_unpack leaves)."""
        self.local.batch_finish(n_samples, gathered_t, self.world if n_ranks is None else int(n_ranks), counts_t[0], counts_t[1],
                                counts_t[2], slot)
        return counts_t

    # the subset words of a block in compact form (include/yacht_hip.h: yh_run_batch_words_*)
    def batch_words_pack(self, words_t, packed_t, cap: int):
        self.local.words_pack(words_t, packed_t, cap)

    def batch_words_unpack(self, gathered_t, cap: int, words_out_t, overflow_t):
        self.local.words_unpack(gathered_t, self.world if self.has_exchange else 1, cap, words_out_t, overflow_t)

    def run_batch(self, samples, counts_t=None):
        """Up to 256 samples in one pass: this rank's share of the [3, B, N_total] counts (sum them with reduce())."""
        import torch

        B = len(samples)
        assert 1 <= B <= 256
        if counts_t is None:
            counts_t = torch.zeros((3, B, self.n_total), dtype=torch.int32, device=self.dev)
        nw = batch_planes(B) * self.n_total
        words = torch.zeros(nw, dtype=torch.int64, device=self.dev)
        gathered = torch.zeros((self.world, nw), dtype=torch.int64, device=self.dev)
        self.batch_begin(self.pack_batch(samples), counts_t, words)
        self.batch_exchange(words, gathered)
        return self.batch_end(B, gathered, counts_t)

    def reduce(self, counts_t, dst=None):
        """Sum of the ranks' shares: the [.., 3, N_total] counts of the whole database (on `dst`, or on every rank)."""
        import torch.distributed as dist

        if self.world == 1 and not self.has_exchange:
            return counts_t
        c = _stage(counts_t, self.group).contiguous()
        if dst is None:
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dist.reduce(c, dst=dst, op=dist.ReduceOp.SUM, group=self.group)
        return c.to(self.dev)

    def gather(self, counts_t):
        """(the name ShardedRefDB uses for 'the whole table on every rank')"""
        return self.reduce(counts_t.clone())

    def close(self):
        self.local.close()


# ======================================================================================================
# The result path of the batched hash-range run: compact rows instead of dense [3, B, N] shares
# ======================================================================================================
# north_star: "only a final RCCL gather of per-reference overlap counts".  A block of B samples leaves every rank with
# three dense [B, N] shares, almost all zero (a 10^6-hash sample overlaps a few hundred of 85 205 references): summing
# them as they are moves 3 * B * N * 4 bytes per rank and block (65.4 MB at B = 64, rs214 scale) -- more than the
# rank's compute for the block takes.  After the second half every rank knows the GLOBAL subset (the OR of the gathered
# words), so every rank has the SAME entries -- one per set bit s of word r, in (r, s) order -- and only the values
# differ: a rank packs its (overlap, n_excl, n_match) share per entry (yh_run_batch_rows_pack_device), ONE sum-reduce of
# the first `cap` value triples (12 bytes per entry: no keys on the wire) gives the totals, and the consumer unpacks the
# rows (sample, ref, overlap, n_excl, n_match).  `cap` -- the size of the collective -- must be the same on every rank
# and known on the host before the collective is queued, while the number of entries is only known on the device: the
# collective of a block takes the current cap; when the block's entry count (the same number on every rank, read back
# behind the block -- finish() -- at the same point of the program on every rank) exceeds it, all ranks fall back to
# the dense reduce for THAT block (its dense shares are still in place) and raise the cap for the following ones.
class BatchRowsReducer:
    """Sum of the ranks' shares of a batch, as compact rows.  One object per HashRangeRefDB; `nbuf` blocks in flight.

    send(b, n_samples, counts_t, slot)   behind hr.batch_end(.., slot): pack + ONE reduce of cap * 12 bytes (asynchronous
                                         on RCCL)
    finish(b) -> (rows, dense)           rows: [n, 5] int32 tensor (sample, ref, overlap, n_excl, n_match) on `dst` (on
                                         every rank when dst is None), None elsewhere -- a view of buffer b's own row
                                         array, untouched until the next finish(b) of the same b; dense: the summed
                                         [3, B, N] block instead, when the block had more entries than the collective carried
    Call finish(b) before the batch slot and the counts of block b are used again."""

    def __init__(self, hr: "HashRangeRefDB", batch: int, dst: Optional[int] = 0, nbuf: int = 3, cap_rows: Optional[int] = None):
        import torch

        self.hr, self.dst, self.nbuf = hr, dst, int(nbuf)
        self.B = int(batch)
        self.cap = int(cap_rows) if cap_rows else min(384 * self.B, self.B * hr.n_total)  # (B * N entries at most)
        self.dev = hr.dev
        self._alloc = 0
        self.vals = [None] * self.nbuf
        self.rows = [None] * self.nbuf
        self.nrows_dev = [torch.zeros(1, dtype=torch.int32, device=self.dev) for _ in range(self.nbuf)]
        pin = self.dev.type == "cuda"
        self.nrows_host = [torch.zeros(1, dtype=torch.int32).pin_memory() if pin else torch.zeros(1, dtype=torch.int32)
                           for _ in range(self.nbuf)]
        self.ev = [torch.cuda.Event() if pin else None for _ in range(self.nbuf)]
        self.state = [None] * self.nbuf  # (n_samples, counts_t, slot, cap used, pending work)
        self.bytes_sent = 0       # what this rank put into the result collectives so far
        self.bytes_dense = 0      # ... what the dense form would have been
        self.n_overflow = 0
        self._grow()

    def _grow(self):
        import torch

        if self._alloc >= self.cap:
            return
        self._alloc = self.cap
        for b in range(self.nbuf):
            if self.state[b] is None:
                self.vals[b] = torch.zeros((self._alloc, 3), dtype=torch.int32, device=self.dev)
        # one row buffer per in-flight block (a caller may still be reading block b's rows -- e.g. through a
        # non-blocking D2H copy -- when the next block is finished); a buffer in use keeps its old size until it is free
        for b in range(self.nbuf):
            if self.state[b] is None:
                self.rows[b] = torch.zeros((self._alloc, 5), dtype=torch.int32, device=self.dev)
        self._nrows_unpack = torch.zeros(1, dtype=torch.int32, device=self.dev)

    def _is_dst(self) -> bool:
        return self.dst is None or self.hr.rank == self.dst

    def send(self, b: int, n_samples: int, counts_t, slot: int = 0):
        import torch.distributed as dist

        hr = self.hr
        assert self.state[b] is None, "finish(b) first"
        if self.vals[b] is None or self.vals[b].shape[0] < self.cap:
            self._grow()
            import torch

            self.vals[b] = torch.zeros((self._alloc, 3), dtype=torch.int32, device=self.dev)
        cap = self.cap
        vals = self.vals[b][:cap]
        hr.local.rows_pack(counts_t, vals, self.nrows_dev[b], slot)
        self.nrows_host[b].copy_(self.nrows_dev[b], non_blocking=True)
        if self.ev[b] is not None:
            self.ev[b].record()
        work = None
        if hr.has_exchange:
            if vals.is_cuda and not _is_gloo(hr.group):
                work = (dist.all_reduce(vals, op=dist.ReduceOp.SUM, group=hr.group, async_op=True) if self.dst is None else
                        dist.reduce(vals, dst=self.dst, op=dist.ReduceOp.SUM, group=hr.group, async_op=True))
            else:
                c = _stage(vals, hr.group).contiguous()
                if self.dst is None:
                    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=hr.group)
                else:
                    dist.reduce(c, dst=self.dst, op=dist.ReduceOp.SUM, group=hr.group)
                vals.copy_(c)
        self.bytes_sent += cap * 12
        self.bytes_dense += 3 * n_samples * hr.n_total * 4
        self.state[b] = (n_samples, counts_t, slot, cap, work)

    def finish(self, b: int):
        n_samples, counts_t, slot, cap, work = self.state[b]
        self.state[b] = None
        if work is not None:
            work.wait()
        if self.ev[b] is not None:
            self.ev[b].synchronize()
        n = int(self.nrows_host[b].item())  # (the same number on every rank: the global subset's size)
        self.last_n_rows = n
        if n > cap:  # the collective was too small for this block: its dense shares are still there -- all ranks take this branch
            self.n_overflow += 1
            self.cap = min(((n + n // 4 + 4095) // 4096) * 4096, max(self.B * self.hr.n_total, n))
            dense = self.hr.reduce(counts_t, dst=self.dst)
            self.bytes_sent += 3 * n_samples * self.hr.n_total * 4
            return None, (dense if self._is_dst() else None)
        if not self._is_dst():
            return None, None
        if self.rows[b] is None or self.rows[b].shape[0] < cap:
            import torch

            self.rows[b] = torch.zeros((max(self._alloc, cap), 5), dtype=torch.int32, device=self.dev)
        self.hr.local.rows_unpack(self.vals[b][:cap], self.rows[b][:cap], self._nrows_unpack, slot)
        return self.rows[b][:n], None  # (block b's own buffer: valid until finish(b) of a LATER block in the same buffer)

    @staticmethod
    def rows_to_dense(rows_t, n_samples: int, n_refs: int):
        """[3, B, N] int32 from the compact rows (tests, parity checks)."""
        import torch

        out = torch.zeros((3, n_samples, n_refs), dtype=torch.int32, device=rows_t.device)
        if rows_t.shape[0]:
            s_, r_ = rows_t[:, 0].long(), rows_t[:, 1].long()
            for k in range(3):
                out[k, s_, r_] = rows_t[:, 2 + k]
        return out


# ======================================================================================================
# The batched hash-range run as ONE object: blocks of <= 256 samples through three batch slots
# ======================================================================================================
def words_packed_len(cap: int) -> int:
    """int64 words of a packed subset-word buffer of capacity `cap` (include/yacht_hip.h: yh_run_batch_words_packed_len)."""
    return 1 + int(cap) + (int(cap) + 1) // 2


class BatchedRangeRunner:
    """The throughput form of `yacht run` over hash-range shards, driven block by block.

    Per block (<= 256 distinct samples -- 64 until round 5: VERDICT r05 weak 5, an eighth of 64 samples was too little work per
    block for eight ranks --, packed once with hr.pack_batch): first half (lookups of this rank's slices) -> the
    block's subset words (one per reference and plane of 64 samples) leave in ONE all-gather -- in compact form: a rank's
    NON-ZERO (word, index) entries, 12 * cap_words + 8 bytes instead of 8 * planes * N (a block of 64 samples overlaps
    ~15 000 of 85 205 references) -> second half
    (subset = OR over the ranks, exclusive pass) -> the block's result leaves as compact rows (BatchRowsReducer: ONE
    sum-reduce of 12 * cap_rows bytes).  `nbuf` blocks rotate through the library's batch slots: while the words of block
    j travel, the second half of block j - 1 runs and its rows leave; block j - nbuf's entry count and overflow flag are
    read back before its slot is reused -- the only host synchronisation, one block of queued work behind the GPU.

    Both collectives have a FIXED size that every rank must know before the entry counts exist; a block that did not fit
    (the same verdict on every rank: every rank sees every rank's count) is repeated synchronously -- words with capacity
    N, rows through the dense reduce -- and the capacities grow for the blocks that follow.

    submit(batch, n_samples, tag)  queue a block (batch = hr.pack_batch(..)); may deliver earlier blocks' results
    drain()                        finish everything queued
    Results arrive in submission order through on_result(tag, n_samples, rows, dense): on rank `dst` (every rank when
    dst is None) rows = [n, 5] int32 (sample, ref, overlap, n_excl, n_match) or, after a rows overflow, dense = the
    summed [3, B, N] block; (None, None) elsewhere.  No reference counterpart (run_YACHT.py:150: one sample per process)."""

    def __init__(self, hr: "HashRangeRefDB", batch: int = 64, dst: Optional[int] = 0, nbuf: int = 3, cap_rows: Optional[int] = None,
                 cap_words: Optional[int] = None, compact_words: bool = True, dense_rows: bool = False, on_result=None,
                 async_collectives: bool = True, finish_stream: bool = True):
        import torch

        assert 1 <= batch <= 256 and 1 <= nbuf <= 3, "the library has YH_BATCH_SLOTS = 3 batch slots of <= 256 samples"
        self.hr, self.B, self.nbuf, self.dst = hr, int(batch), int(nbuf), dst
        self.dev = dev = hr.dev
        N = hr.n_total
        self.planes = batch_planes(self.B)
        self.NW = NW = self.planes * N   # subset words of a block: one per reference and plane of 64 samples
        self.compact_words = bool(compact_words)
        self.dense_rows = bool(dense_rows)
        self.cap_words = max(1, min(int(cap_words) if cap_words else 320 * self.B, max(NW, 1)))
        self.on_result = on_result
        self.async_collectives = bool(async_collectives) and dev.type == "cuda" and hr.has_exchange and not _is_gloo(hr.group)
        self.red = None if self.dense_rows else BatchRowsReducer(hr, batch=self.B, dst=dst, nbuf=self.nbuf, cap_rows=cap_rows)
        self.counts = [torch.zeros((3, self.B, N), dtype=torch.int32, device=dev) for _ in range(self.nbuf)]
        self.words = [torch.zeros(NW, dtype=torch.int64, device=dev) for _ in range(self.nbuf)]
        self.words_or = [torch.zeros((1, NW), dtype=torch.int64, device=dev) for _ in range(self.nbuf)]
        self.gath_dense = [None] * self.nbuf   # [world, planes * N] int64, made on first use (dense exchange only)
        self.packed = [None] * self.nbuf       # this rank's packed words / the all-gathered ones, sized for the cap in use
        self.gath_packed = [None] * self.nbuf
        self.ovf_dev = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(self.nbuf)]
        pin = dev.type == "cuda"
        self.ovf_host = [torch.zeros(1, dtype=torch.int32).pin_memory() if pin else torch.zeros(1, dtype=torch.int32)
                         for _ in range(self.nbuf)]
        self.ev = [torch.cuda.Event() if pin else None for _ in range(self.nbuf)]
        # The second halves on a stream of their own (include/yacht_hip.h: yh_db_set_batch_finish_stream): a dozen launches at
        # their launch floors (~100 us a block at rs214 scale, whatever the rank's share of the lookups) run beside the next
        # block's lookups instead of between them.  ev_x[b]: block b's words are on their way (recorded on the first stream).
        self.s2 = None
        if finish_stream and pin and hasattr(hr.local, "set_finish_stream"):
            self.s2 = torch.cuda.Stream(device=dev)
            hr.local.set_finish_stream(self.s2)
        self.ev_x = [torch.cuda.Event() if self.s2 is not None else None for _ in range(self.nbuf)]
        self.inflight = [None] * self.nbuf     # (seq, tag, n_samples, batch, pending dense reduce)
        self.prev = None                       # first half queued, words travelling: (seq, tag, n_samples, batch, work, cap)
        self.seq = 0
        self.n_words_overflow = 0
        self.bytes_words = 0                   # what this rank put into the word exchanges so far
        self.last_result = None

    # ---- what a block puts on the wire ----------------------------------------------------------------
    def collective_bytes(self) -> dict:
        N = self.hr.n_total
        w = 8 * words_packed_len(self.cap_words) if self.compact_words else 8 * self.NW
        r = 3 * self.B * N * 4 if self.red is None else 12 * self.red.cap
        return {"subset_words_all_gather": w, "subset_words_dense_form": 8 * self.NW, "subset_words_capacity": self.cap_words if self.compact_words else None,
                "result": r, "result_dense_form": 3 * self.B * N * 4, "result_capacity_rows": None if self.red is None else self.red.cap,
                "total": w + r}

    def _exchange(self, b: int, cap: int, sync: bool = False):
        """The block's subset words on their way (their pack included).  With a finish stream the library leaves a first half's
        words THERE (the first stream carries nothing but clears and lookups), so the pack and the collective are issued there."""
        if self.s2 is None:
            return self._exchange_on(b, cap, sync)
        import torch

        with torch.cuda.stream(self.s2):
            return self._exchange_on(b, cap, sync)

    def _exchange_on(self, b: int, cap: int, sync: bool = False):
        import torch

        hr = self.hr
        if not self.compact_words:
            if self.gath_dense[b] is None:
                self.gath_dense[b] = torch.zeros((hr.world, self.NW), dtype=torch.int64, device=self.dev)
            self.bytes_words += 8 * self.NW
            work = hr.batch_exchange(self.words[b], self.gath_dense[b], async_op=self.async_collectives and not sync)
            if self.s2 is not None:
                self.ev_x[b].record()
            return work
        L = words_packed_len(cap)
        if self.packed[b] is None or self.packed[b].numel() != L:
            self.packed[b] = torch.zeros(L, dtype=torch.int64, device=self.dev)
            self.gath_packed[b] = torch.zeros((hr.world, L), dtype=torch.int64, device=self.dev)
        hr.batch_words_pack(self.words[b], self.packed[b], cap)
        self.bytes_words += 8 * L
        work = hr.batch_exchange(self.packed[b], self.gath_packed[b], async_op=self.async_collectives and not sync)
        if self.s2 is not None:
            self.ev_x[b].record()  # (behind the pack, a copy that stands in for the exchange, a synchronous collective -- on the stream they were issued on)
        return work

    def close(self):
        """Everything queued finished, the handle back on one stream (the runner's second stream is not used after this)."""
        self.drain()
        if self.s2 is not None:
            self.s2.synchronize()
            self.hr.local.set_finish_stream(None)
            self.s2 = None

    def _second_half(self, seq, tag, n_in, batch, work, cap):
        if self.s2 is None:
            return self._second_half_on(seq, tag, n_in, batch, work, cap)
        import torch

        with torch.cuda.stream(self.s2):  # (the library's side of these calls runs on s2 by itself; this is for torch's side)
            self.s2.wait_event(self.ev_x[seq % self.nbuf])
            return self._second_half_on(seq, tag, n_in, batch, work, cap)

    def _second_half_on(self, seq, tag, n_in, batch, work, cap):
        hr = self.hr
        b = seq % self.nbuf
        if work is not None:
            work.wait()
        if self.compact_words:
            hr.batch_words_unpack(self.gath_packed[b], cap, self.words_or[b], self.ovf_dev[b])
            hr.batch_end(n_in, self.words_or[b], self.counts[b], slot=b, n_ranks=1)
            self.ovf_host[b].copy_(self.ovf_dev[b], non_blocking=True)
        else:
            hr.batch_end(n_in, self.gath_dense[b], self.counts[b], slot=b)
        pend = None
        if self.red is not None:
            self.red.send(b, n_in, self.counts[b], slot=b)  # (its event, recorded behind the count's copy, covers the flag's copy too)
        else:
            import torch.distributed as dist

            if hr.has_exchange:
                if self.async_collectives:
                    pend = dist.reduce(self.counts[b], dst=self.dst, op=dist.ReduceOp.SUM, group=hr.group, async_op=True) if self.dst is not None \
                        else dist.all_reduce(self.counts[b], op=dist.ReduceOp.SUM, group=hr.group, async_op=True)
                else:
                    self.counts[b].copy_(hr.reduce(self.counts[b], dst=self.dst))
            # The block's event goes BEHIND the reduce: a synchronous reduce is queued on this stream (the finish stream when
            # there is one), and _finish hands counts[b] to on_result -- and the slot's next first half clears it -- once this
            # event has passed (recorded in front of the reduce, both raced with it).  An asynchronous reduce is
            # ordered by pend.wait() in _finish.
            if self.ev[b] is not None:
                self.ev[b].record()
        self.inflight[b] = (seq, tag, n_in, batch, pend)

    def _is_dst(self) -> bool:
        return self.dst is None or self.hr.rank == self.dst

    def _finish(self, b: int):
        if self.inflight[b] is None:
            return
        seq, tag, n_in, batch, pend = self.inflight[b]
        self.inflight[b] = None
        rows = dense = None
        if self.red is not None:
            rows, dense = self.red.finish(b)  # (waits for the block's event: the overflow flag has landed too)
        else:
            if pend is not None:
                pend.wait()
            if self.ev[b] is not None:
                self.ev[b].synchronize()
            dense = self.counts[b] if self._is_dst() else None
        if self.compact_words and int(self.ovf_host[b].item()) != 0:
            # Some rank had more non-zero words than the exchange carried: the subset of this block was incomplete on every
            # rank, and every rank knows (the flag is made from all ranks' counts).  Repeat the block here, synchronously,
            # with a capacity nothing can exceed, and give the following blocks a larger one.
            self.n_words_overflow += 1
            N = max(self.NW, 1)
            self.cap_words = min(N, max(2 * self.cap_words, 1024))
            rows, dense = self._redo(b, n_in, batch, N)
        self.last_result = (tag, n_in, rows, dense)
        if self.on_result is not None:
            self.on_result(tag, n_in, rows, dense)

    def _redo(self, b: int, n_in: int, batch, cap: int):
        hr = self.hr
        hr.batch_begin(batch, self.counts[b], self.words[b], slot=b)
        w = self._exchange(b, cap, sync=True)

        def second():
            if w is not None:
                w.wait()
            hr.batch_words_unpack(self.gath_packed[b], cap, self.words_or[b], self.ovf_dev[b])
            hr.batch_end(n_in, self.words_or[b], self.counts[b], slot=b, n_ranks=1)
            if self.red is not None:
                self.red.send(b, n_in, self.counts[b], slot=b)

        if self.s2 is None:
            second()
        else:
            import torch

            with torch.cuda.stream(self.s2):
                self.s2.wait_event(self.ev_x[b])
                second()
            self.s2.synchronize()  # (the buffers dropped below and the dense reduce on the first stream)
        self.packed[b] = self.gath_packed[b] = None  # (sized for `cap`: the next block makes its own)
        if self.red is not None:
            return self.red.finish(b)
        dense = hr.reduce(self.counts[b], dst=self.dst)
        return None, (dense if self._is_dst() else None)

    def submit(self, batch, n_samples: int, tag=None):
        assert 1 <= n_samples <= self.B
        seq = self.seq
        self.seq += 1
        b = seq % self.nbuf
        if self.nbuf == 1 and self.prev is not None:
            # One slot: the previous block still owns it (first half, counts, gathered words) until its second half has run,
            # so the two halves cannot overlap -- finish it here, in front of this block's first half.
            self._second_half(*self.prev)
            self.prev = None
        self._finish(b)  # block seq - nbuf: its slot, its counts and its buffers are free again
        self.hr.batch_begin(batch, self.counts[b], self.words[b], slot=b)
        cap = self.cap_words
        work = self._exchange(b, cap)
        if self.prev is not None:  # ... and behind this block's first half: the second half of the previous one
            self._second_half(*self.prev)
        self.prev = (seq, tag, n_samples, batch, work, cap)

    def drain(self):
        if self.prev is not None:
            self._second_half(*self.prev)
            self.prev = None
        order = sorted((q for q in range(self.nbuf) if self.inflight[q] is not None), key=lambda q: self.inflight[q][0])
        for b in order:  # (oldest block first: the same order on every rank)
            self._finish(b)
