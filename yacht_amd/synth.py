"""Seeded synthetic FracMinHash sketches shaped like BASELINE.json's configs.

Hashes are uniform in [0, max_hash(scaled)), unique and ascending inside a sketch — the
properties sourmash "mins" arrays have (murmur64 output below the scaled cut-off).  Used by the
parity tests and by bench.py; there is no network, so these stand in for GTDB sketches.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def max_hash_for_scaled(scaled: int) -> int:
    """sourmash's cut-off: round(2**64 / scaled); 18446744073709552 for scaled=1000."""
    return (2 ** 64 + scaled // 2) // scaled if scaled > 1 else 2 ** 64 - 1


def random_sketch(rng: np.random.Generator, size: int, max_hash: int) -> np.ndarray:
    if size <= 0:
        return np.zeros(0, dtype=np.uint64)
    return np.unique(rng.integers(0, max_hash, size=size, dtype=np.uint64))


def lognormal_sizes(rng, n: int, median: float, sigma: float, lo: int, hi: int) -> np.ndarray:
    s = np.rint(rng.lognormal(np.log(median), sigma, size=n)).astype(np.int64)
    return np.clip(s, lo, hi)


def pack(sketches: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    offsets = np.zeros(len(sketches) + 1, dtype=np.uint64)
    if len(sketches):
        offsets[1:] = np.cumsum([len(s) for s in sketches], dtype=np.uint64)
    values = np.concatenate(sketches).astype(np.uint64) if len(sketches) and int(offsets[-1]) else np.zeros(0, np.uint64)
    return values, offsets


def independent_refs(rng, n_refs: int, median: float, sigma: float, lo: int, hi: int, scaled: int = 1000) -> List[np.ndarray]:
    mh = max_hash_for_scaled(scaled)
    return [random_sketch(rng, int(s), mh) for s in lognormal_sizes(rng, n_refs, median, sigma, lo, hi)]


def clustered_refs(rng, n_clusters: int, retentions: Sequence[float], size: int, scaled: int = 1000,
                   private_fill: bool = True) -> List[np.ndarray]:
    """Clusters of len(retentions) genomes: member k keeps a Bernoulli(retentions[k]) subset of the
    cluster parent's hashes and (optionally) tops up to ~`size` with private hashes.  With
    retentions (1, .9, .5, .25, .1) the pairwise containments straddle 0.95**31 = 0.2039."""
    mh = max_hash_for_scaled(scaled)
    out = []
    for _ in range(n_clusters):
        parent = random_sketch(rng, size, mh)
        for r in retentions:
            keep = parent[rng.random(parent.size) < r] if r < 1.0 else parent
            if private_fill and keep.size < size:
                keep = np.union1d(keep, random_sketch(rng, size - keep.size, mh))
            out.append(keep.astype(np.uint64))
    return out


def sample_from_refs(rng, refs: Sequence[np.ndarray], present: Sequence[int], coverage, n_sample: int,
                     scaled: int = 1000) -> np.ndarray:
    """A metagenome-like sample: Bernoulli(coverage[k]) subsample of refs[present[k]] for each
    present genome, plus uniform noise hashes up to ~n_sample distinct hashes."""
    mh = max_hash_for_scaled(scaled)
    parts = []
    cov = np.broadcast_to(np.asarray(coverage, dtype=np.float64), (len(present),))
    for k, j in enumerate(present):
        r = refs[j]
        parts.append(r[rng.random(r.size) < cov[k]])
    have = int(sum(p.size for p in parts))
    if n_sample > have:
        parts.append(rng.integers(0, mh, size=n_sample - have, dtype=np.uint64))
    return np.unique(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)


# ---- the named configurations --------------------------------------------------------------------------
def config2(seed: int = 1001, n_refs: int = 1000, n_sample: int = 1_000_000):
    """BASELINE.json configs[1]: 1 000 refs x ~5 000 hashes vs one ~1 M-hash sample."""
    rng = np.random.default_rng(seed)
    refs = independent_refs(rng, n_refs, 5000, 0.35, 500, 20000)
    present = rng.choice(n_refs, size=min(50, n_refs), replace=False)
    sample = sample_from_refs(rng, refs, present, 0.3, n_sample)
    values, offsets = pack(refs)
    return values, offsets, sample


def config3_like(seed: int = 1002, n_refs: int = 85_205, n_sample: int = 1_000_000, cluster_frac: float = 0.10,
                 n_present: int = 200):
    """BASELINE.json configs[2] shape (GTDB rs214 representatives) at any n_refs: sizes
    LogNormal(ln 3300, 0.6) clipped to [300, 15000]; ~10 % of the genomes sit in clusters of
    2-8 that share 10-95 % of a parent; the sample holds n_present genomes at coverage
    Beta(0.5, 2) plus noise."""
    rng = np.random.default_rng(seed)
    mh = max_hash_for_scaled(1000)
    sizes = lognormal_sizes(rng, n_refs, 3300, 0.6, 300, 15000)
    refs: List[np.ndarray] = []
    j = 0
    while j < n_refs:
        if rng.random() < cluster_frac / 4.0 and j + 2 <= n_refs:  # mean cluster size ~4.x
            k = int(min(rng.integers(2, 9), n_refs - j))
            parent = random_sketch(rng, int(sizes[j]), mh)
            for t in range(k):
                share = rng.uniform(0.10, 0.95)
                keep = parent[rng.random(parent.size) < share]
                fill = max(int(sizes[j + t]) - keep.size, 0)
                refs.append(np.union1d(keep, random_sketch(rng, fill, mh)).astype(np.uint64))
            j += k
        else:
            refs.append(random_sketch(rng, int(sizes[j]), mh))
            j += 1
    present = rng.choice(n_refs, size=min(n_present, n_refs), replace=False)
    cov = rng.beta(0.5, 2.0, size=present.size)
    sample = sample_from_refs(rng, refs, present, cov, n_sample)
    values, offsets = pack(refs)
    return values, offsets, sample


def config4(seed: int = 1003, n_clusters: int = 2000, size: int = 5000):
    """BASELINE.json configs[3]: train pairwise, clusters x 5 with retentions straddling C."""
    rng = np.random.default_rng(seed)
    refs = clustered_refs(rng, n_clusters, (1.0, 0.9, 0.5, 0.25, 0.1), size)
    values, offsets = pack(refs)
    return values, offsets


# ---- device-side generation (torch is plumbing here: it only makes the synthetic input) ----------------
def config3_device(seed: int = 1002, n_refs: int = 85_205, n_sample: int = 1_000_000, device: str = "cuda:0",
                   cluster_frac: float = 0.10, n_present: int = 200, median: float = 3300.0, sigma: float = 0.6,
                   lo: int = 300, hi: int = 15000, scaled: int = 1000):
    """configs[2] (GTDB rs214 representatives scale) generated directly in HBM.

    Returns (values int64[H], offsets int64[N+1], sample int64[|S|]) as torch tensors on
    `device`; all hashes are < 2**63 for scaled >= 2, so the int64 bit patterns ARE the uint64
    hashes.  Same distribution family as config3_like (sizes LogNormal, ~cluster_frac of the
    genomes in clusters of 2-8 sharing 10-95 % of a parent, sample = n_present genomes at
    coverage Beta(0.5, 2) + uniform noise), different random stream.
    """
    import torch

    assert scaled >= 2
    mh = max_hash_for_scaled(scaled)
    rng = np.random.default_rng(seed)
    sizes = lognormal_sizes(rng, n_refs, median, sigma, lo, hi)
    # cluster structure on the host (N-sized): parent[j] = j for founders / singletons
    parent = np.arange(n_refs, dtype=np.int64)
    share = np.zeros(n_refs, dtype=np.float64)
    j = 0
    while j < n_refs:
        if rng.random() < cluster_frac / 4.0 and j + 2 <= n_refs:
            k = int(min(rng.integers(2, 9), n_refs - j))
            parent[j + 1 : j + k] = j
            share[j + 1 : j + k] = rng.uniform(0.10, 0.95, size=k - 1)
            j += k
        else:
            j += 1
    offsets_np = np.zeros(n_refs + 1, dtype=np.int64)
    offsets_np[1:] = np.cumsum(sizes)
    H = int(offsets_np[-1])

    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    offsets = torch.from_numpy(offsets_np).to(dev)
    sizes_t = torch.from_numpy(sizes).to(dev)
    seg = torch.repeat_interleave(torch.arange(n_refs, device=dev, dtype=torch.int32), sizes_t)
    vals = torch.randint(0, mh, (H,), generator=g, device=dev, dtype=torch.int64)
    # cluster members copy the founder's hash at the same slot with probability share[j]
    par_t = torch.from_numpy(parent).to(dev)
    share_t = torch.from_numpy(share).to(dev)
    seg64 = seg.long()
    pos = torch.arange(H, device=dev, dtype=torch.int64) - offsets[seg64]
    pj = par_t[seg64]
    take = (torch.rand(H, generator=g, device=dev, dtype=torch.float64) < share_t[seg64]) & (pos < sizes_t[pj])
    src = offsets[pj] + pos
    vals = torch.where(take, vals[src.clamp_(0, H - 1)], vals)
    del pos, pj, take, src, seg64
    # order: by hash, then stably by reference -> each reference's slice ascending
    vals, perm = torch.sort(vals)
    seg = seg[perm]
    del perm
    seg, perm2 = torch.sort(seg, stable=True)
    vals = vals[perm2]
    del perm2
    # (astronomically rare) equal neighbours inside one reference: nudge the second one up
    dup = (vals[1:] == vals[:-1]) & (seg[1:] == seg[:-1])
    if bool(dup.any()):
        vals[1:] += dup.long()
    # sample
    present = rng.choice(n_refs, size=min(n_present, n_refs), replace=False)
    cov = np.zeros(n_refs, dtype=np.float64)
    cov[present] = rng.beta(0.5, 2.0, size=present.size)
    cov_t = torch.from_numpy(cov).to(dev)
    keep = torch.rand(H, generator=g, device=dev, dtype=torch.float64) < cov_t[seg.long()]
    picked = vals[keep]
    del keep, seg
    n_noise = max(n_sample - int(picked.numel()), 0)
    noise = torch.randint(0, mh, (n_noise,), generator=g, device=dev, dtype=torch.int64)
    sample = torch.unique(torch.cat([picked, noise]))  # sorted ascending, distinct
    return vals.contiguous(), offsets.contiguous(), sample.contiguous()
