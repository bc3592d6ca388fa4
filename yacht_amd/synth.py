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
    del seg
    vals = vals.contiguous()
    offsets = offsets.contiguous()
    sample = sample_device(vals, offsets, seed=seed, n_sample=n_sample, n_present=n_present, scaled=scaled, _rng=rng, _gen=g)
    return vals, offsets, sample


def sample_device(values, offsets, seed: int, n_sample: int = 1_000_000, n_present: int = 200, scaled: int = 1000,
                  shape: str = "present", _rng=None, _gen=None):
    """One more sample sketch for a database that already lives in HBM (int64 tensors as config3_device
    returns them), sorted ascending and distinct.

    shape = "present": n_present genomes at coverage Beta(0.5, 2) + uniform noise up to n_sample hashes
            (cfg 3's sample).
    shape = "real":    the hit shape of the reference's shipped results (SURVEY.md 6:
            use_case_examples/low_abundance_samples -- ~25 000 of 85 205 references overlap an 83 k-hash
            sample): 29 % of the references get 1 + Geometric(0.32) hashes each (capped at 50, mean ~3),
            noise tops the sample up to n_sample.  Nearly every sample hash is a database hash, and
            ~30 % of the references are in the exclusive-count subset."""
    import torch

    dev = values.device
    n_refs = int(offsets.numel() - 1)
    H = int(values.numel())
    mh = max_hash_for_scaled(scaled)
    rng = _rng if _rng is not None else np.random.default_rng(seed)
    g = _gen
    if g is None:
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
    sizes = (offsets[1:] - offsets[:-1])
    cov = np.zeros(n_refs, dtype=np.float64)
    if shape == "present":
        present = rng.choice(n_refs, size=min(n_present, n_refs), replace=False)
        cov[present] = rng.beta(0.5, 2.0, size=present.size)
    elif shape == "real":
        k = max(int(round(0.29 * n_refs)), 1)
        present = rng.choice(n_refs, size=k, replace=False)
        want = np.minimum(rng.geometric(0.32, size=k), 50).astype(np.float64)
        cov[present] = want / np.maximum(sizes.cpu().numpy()[present], 1)
    else:
        raise ValueError(shape)
    cov_t = torch.from_numpy(cov).to(dev)
    seg = torch.repeat_interleave(torch.arange(n_refs, device=dev, dtype=torch.int64), sizes)
    keep = torch.rand(H, generator=g, device=dev, dtype=torch.float64) < cov_t[seg]
    del seg
    picked = values[keep]
    del keep
    n_noise = max(n_sample - int(picked.numel()), 0)
    noise = torch.randint(0, mh, (n_noise,), generator=g, device=dev, dtype=torch.int64)
    return torch.unique(torch.cat([picked, noise])).contiguous()  # sorted ascending, distinct


def real_shape_sample(rng, refs: Sequence[np.ndarray], n_sample: int, frac_overlapping: float = 0.29,
                      scaled: int = 1000) -> np.ndarray:
    """Host twin of sample_device(shape="real") for oracle-sized parity cases."""
    mh = max_hash_for_scaled(scaled)
    n_refs = len(refs)
    k = max(int(round(frac_overlapping * n_refs)), 1)
    present = rng.choice(n_refs, size=k, replace=False)
    parts = []
    for j in present:
        r = refs[int(j)]
        if r.size:
            c = min(int(min(rng.geometric(0.32), 50)), r.size)
            parts.append(rng.choice(r, size=c, replace=False))
    have = int(sum(p.size for p in parts))
    if n_sample > have:
        parts.append(rng.integers(0, mh, size=n_sample - have, dtype=np.uint64))
    return np.unique(np.concatenate(parts)).astype(np.uint64) if parts else np.zeros(0, np.uint64)


# ---- one global database any rank can generate a slice of (bench.py --gpus N, tests) ---------------------
# Counter-based: hash k of reference j is a pure function of (seed, j, k), so the shards of ONE
# database -- clusters straddling the cuts included -- are generated on their own GPUs with no exchange,
# and any rank can regenerate any reference (the sample's present genomes; the parity check on rank 0).
_M64 = (1 << 64) - 1


def _i64(x: int) -> int:
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


def _mix64_t(z):
    """splitmix64 finaliser on int64 tensors (wrapping multiply, logical shifts)."""
    z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * _i64(0x94D049BB133111EB)
    return z ^ ((z >> 31) & ((1 << 33) - 1))


def global_db_plan(seed: int, n_refs: int, cluster_frac: float = 0.10, median: float = 3300.0, sigma: float = 0.6,
                   lo: int = 300, hi: int = 15000):
    """Host-side structure of the global database (identical on every rank): sizes, cluster founder of
    every reference (itself for singletons), the share of the founder's hashes a member copies."""
    rng = np.random.default_rng(seed)
    sizes = lognormal_sizes(rng, n_refs, median, sigma, lo, hi)
    parent = np.arange(n_refs, dtype=np.int64)
    share = np.zeros(n_refs, dtype=np.float64)
    j = 0
    while j < n_refs:
        if rng.random() < cluster_frac / 4.0 and j + 2 <= n_refs:
            k = int(min(rng.integers(2, 9), n_refs - j))
            parent[j + 1: j + k] = j
            share[j + 1: j + k] = rng.uniform(0.10, 0.95, size=k - 1)
            j += k
        else:
            j += 1
    offsets = np.zeros(n_refs + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(sizes)
    return {"seed": seed, "n_refs": n_refs, "sizes": sizes, "parent": parent, "share": share, "offsets": offsets}


def global_db_refs_device(plan, ref_ids, device: str = "cuda:0", scaled: int = 1000):
    """CSR (values int64[H'], offsets int64[len(ref_ids)+1]) of the listed references of the global
    database, every slice ascending and distinct.  `ref_ids`: ascending int64 array (a contiguous
    shard, or the genomes present in a sample)."""
    import torch

    dev = torch.device(device)
    mh = max_hash_for_scaled(scaled)
    ref_ids = np.asarray(ref_ids, dtype=np.int64)
    sizes = plan["sizes"][ref_ids]
    n = int(ref_ids.size)
    offs_np = np.zeros(n + 1, dtype=np.int64)
    offs_np[1:] = np.cumsum(sizes)
    H = int(offs_np[-1])
    offsets = torch.from_numpy(offs_np).to(dev)
    if H == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev), offsets
    sizes_t = torch.from_numpy(sizes).to(dev)
    seg = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), sizes_t)
    slot = torch.arange(H, device=dev, dtype=torch.int64) - offsets[seg]
    gid = torch.from_numpy(ref_ids).to(dev)[seg]
    par = torch.from_numpy(plan["parent"][ref_ids]).to(dev)[seg]
    psize = torch.from_numpy(plan["sizes"][plan["parent"][ref_ids]]).to(dev)[seg]
    shr = torch.from_numpy(plan["share"][ref_ids]).to(dev)[seg]
    s0 = _i64(int(plan["seed"]) * 0x9E3779B97F4A7C15)

    def h(ref, k, salt):
        return _mix64_t(_mix64_t(ref * _i64(0xD1B54A32D192ED03) + s0 + salt) + k * _i64(0x9E3779B97F4A7C15))

    own = (h(gid, slot, 1) & ((1 << 62) - 1)) % mh
    u = (h(gid, slot, 2) & ((1 << 53) - 1)).to(torch.float64) / float(1 << 53)
    take = (u < shr) & (slot < psize) & (par != gid)
    founder = (h(par, slot, 1) & ((1 << 62) - 1)) % mh
    vals = torch.where(take, founder, own)
    del own, u, take, founder, gid, par, psize, shr, slot
    vals, perm = torch.sort(vals)
    seg = seg[perm]
    del perm
    seg, perm2 = torch.sort(seg, stable=True)
    vals = vals[perm2]
    del perm2
    # (rare) equal neighbours inside one reference: nudge later ones up until the slice is strictly ascending
    for _ in range(4):
        dup = (vals[1:] <= vals[:-1]) & (seg[1:] == seg[:-1])
        if not bool(dup.any()):
            break
        vals[1:] += dup.long()
    return vals.contiguous(), offsets.contiguous()


def global_db_sample_device(plan, seed: int, n_sample: int = 1_000_000, n_present: int = 200, device: str = "cuda:0",
                            scaled: int = 1000, shape: str = "present"):
    """A sample sketch against the global database (same on every rank for the same arguments):
    "present": n_present genomes at coverage Beta(0.5, 2) + uniform noise up to n_sample distinct
    hashes; "real": 29 % of the references with 1 + Geometric(0.32) hashes each, capped at 50
    (sample_device's shapes)."""
    import torch

    dev = torch.device(device)
    mh = max_hash_for_scaled(scaled)
    rng = np.random.default_rng(seed)
    n_refs = plan["n_refs"]
    if shape == "present":
        present = np.sort(rng.choice(n_refs, size=min(n_present, n_refs), replace=False))
        cov = rng.beta(0.5, 2.0, size=present.size)
    elif shape == "real":
        k = max(int(round(0.29 * n_refs)), 1)
        present = np.sort(rng.choice(n_refs, size=k, replace=False))
        want = np.minimum(rng.geometric(0.32, size=k), 50).astype(np.float64)
        cov = want / np.maximum(plan["sizes"][present], 1)
    else:
        raise ValueError(shape)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    picked = []
    CH = 4096  # references per generated chunk
    for c0 in range(0, present.size, CH):
        ids = present[c0:c0 + CH]
        v, o = global_db_refs_device(plan, ids, device=device, scaled=scaled)
        seg = torch.repeat_interleave(torch.arange(ids.size, device=dev, dtype=torch.int64), o[1:] - o[:-1])
        keep = torch.rand(v.numel(), generator=g, device=dev, dtype=torch.float64) < torch.from_numpy(cov[c0:c0 + CH]).to(dev)[seg]
        picked.append(v[keep])
    picked = torch.cat(picked) if picked else torch.zeros(0, dtype=torch.int64, device=dev)
    n_noise = max(n_sample - int(picked.numel()), 0)
    noise = torch.randint(0, mh, (n_noise,), generator=g, device=dev, dtype=torch.int64)
    return torch.unique(torch.cat([picked, noise])).contiguous()
