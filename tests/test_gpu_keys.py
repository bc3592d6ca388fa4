"""Both lookups work on pieces of a hash and confirm against the rest: the streaming kernel compares truncated keys
(hash >> stream_shift) and confirms a candidate by the low 32 bits of its position's record, the sample-driven one
picks a bucket by the high bits and compares low words.  These cases are built to make the pieces collide: sample and
database hashes that differ only in low bits, runs of equal truncated keys, full-range (scaled = 1) hashes.  Every case
through BOTH kernels, forced (the library's own choice would take one of them), against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle
from yacht_amd import _lib
from yacht_amd.engine import RefDB, pack_csr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _colliding_case(seed=7):
    """Hashes below 2**54, 512 partitions -> pshift 45, kshift 13: the low 13 bits are not in the key."""
    rng = np.random.default_rng(seed)
    top = 1 << 54
    refs = [np.unique(rng.integers(0, top, size=int(rng.integers(500, 3000)), dtype=np.uint64)) for _ in range(300)]
    flat = np.concatenate(refs)
    picked = rng.choice(flat, size=6000, replace=False)
    low = np.uint64((1 << 13) - 1)
    sample_parts = [
        picked[:2000],                                                   # true members
        picked[2000:4000] ^ np.uint64(1),                                # same key, other hash
        (picked[4000:6000] & ~low) | rng.integers(0, 1 << 13, size=2000, dtype=np.uint64),
        rng.integers(0, top, size=20000, dtype=np.uint64),               # noise
    ]
    # references that hold near-duplicates of sample hashes (same key, not the same hash)
    refs.append(np.unique(picked[:1500] ^ np.uint64(2)))
    refs.append(np.unique(np.concatenate([picked[100:400], picked[2000:2300] ^ np.uint64(1)])))  # some real ones

    # one partition (p = 7) with > 4094 sample hashes and a run of equal keys across the tile boundary
    base = np.uint64(7) << np.uint64(45)
    others = np.unique(np.concatenate(sample_parts))
    below = int(np.count_nonzero((others >= base) & (others < base + np.uint64(1 << 44))))
    dense = base + np.arange(4086 - below, dtype=np.uint64) * np.uint64(1 << 20) + np.uint64(12345)
    run = base + np.uint64(1 << 44) + np.arange(20, dtype=np.uint64)       # 20 hashes, one key
    after = base + np.uint64(1 << 44) + np.arange(1, 1500, dtype=np.uint64) * np.uint64(1 << 20)
    sample_parts += [dense, run, after]
    refs.append(np.unique(np.concatenate([run[[0, 5, 19]], base + np.uint64(1 << 44) + np.array([25, 26], dtype=np.uint64),
                                          dense[::7], after[::3]])))
    refs.append(np.unique(np.concatenate([run[[5, 6, 7]], dense[::11]])))
    refs.append(np.unique(base + np.uint64(1 << 44) + np.arange(20, 60, dtype=np.uint64)))  # key matches, no hash does
    sample = np.unique(np.concatenate(sample_parts))
    # the run of 20 equal keys must start before slot 4094 of partition 7's slice and end after it
    first = int(np.searchsorted(sample, run[0]) - np.searchsorted(sample, base))
    assert first < 4094 <= first + 19, first
    values, offsets = pack_csr(refs)
    return values, offsets, sample


def _check(values, offsets, sample):
    w_ov = oracle.overlap(values, offsets, sample)
    mask = (w_ov > 0).astype(np.uint8)
    w_e, w_m = oracle.exclusive(values, offsets, mask, sample)
    with RefDB(values, offsets) as db:
        info = db.info()
        for mode in (_lib.YH_LOOKUP_STREAM, _lib.YH_LOOKUP_INDEXED):
            db.set_lookup(mode)
            assert db.lookup_choice(sample.size) == mode  # the kernel this pass is named for really runs
            ov, e, m = db.run_counts(sample)
            assert np.array_equal(ov, w_ov), mode
            assert np.array_equal(e, np.where(mask, w_e, 0)), mode
            assert np.array_equal(m, np.where(mask, w_m, 0)), mode
            assert np.array_equal(db.overlap(sample), w_ov), mode
    return info


def test_key_collisions_are_not_hits(hip_lib):
    for seed in (7, 8):
        values, offsets, sample = _colliding_case(seed=seed)
        info = _check(values, offsets, sample)
        assert info["stream_shift"] > 0  # the stream really drops low bits of these hashes


def test_full_range_hashes(hip_lib):
    """scaled = 1 sketches: hashes up to 2**64 - 1; the stream keeps bits 32.. only."""
    rng = np.random.default_rng(3)
    refs = [np.unique(rng.integers(0, 2**64 - 1, size=800, dtype=np.uint64, endpoint=True)) for _ in range(40)]
    refs.append(np.array([0, 1, 2**32, 2**32 + 1, 2**63, 2**64 - 2, 2**64 - 1], dtype=np.uint64))
    flat = np.concatenate(refs)
    sample = np.unique(np.concatenate([rng.choice(flat, 3000), rng.choice(flat, 3000) ^ np.uint64(1 << 5),
                                       np.array([0, 2**32 + 1, 2**64 - 1], dtype=np.uint64)]))
    values, offsets = pack_csr(refs)
    _check(values, offsets, sample)


def _hot_database(seed, n_refs, size, n_hot, hot_lo, hot_hi, n_clusters=300):
    """n_refs sketches of ~size uniform hashes; the first 5 * n_clusters form clusters of five around the train threshold
    (config4's retentions); n_hot "conserved k-mers", each put into hot_lo .. hot_hi randomly chosen sketches."""
    from yacht_amd import synth

    rng = np.random.default_rng(seed)
    mh = synth.max_hash_for_scaled(1000)
    refs = []
    for _ in range(n_clusters):
        parent = np.unique(rng.integers(1, mh, size=size, dtype=np.uint64))
        for keep in (1.0, 0.9, 0.5, 0.25, 0.1):
            kept = parent[rng.random(parent.size) < keep]
            extra = np.unique(rng.integers(1, mh, size=max(size - kept.size, 0), dtype=np.uint64))
            refs.append(np.union1d(kept, extra))
    while len(refs) < n_refs:
        refs.append(np.unique(rng.integers(1, mh, size=int(rng.integers(size // 2, size * 2)), dtype=np.uint64)))
    hot = np.unique(rng.integers(1, mh, size=n_hot, dtype=np.uint64))
    members = [[] for _ in range(n_refs)]
    holders = []
    for h in hot:
        m = int(rng.integers(hot_lo, hot_hi + 1))
        who = rng.choice(n_refs, size=m, replace=False)
        holders.append(m)
        for r in who:
            members[int(r)].append(h)
    refs = [np.union1d(r, np.asarray(m, dtype=np.uint64)) if m else r for r, m in zip(refs, members)]
    return refs, hot, holders


def test_hot_kmers_spill_their_buckets_and_nothing_else(hip_lib):
    """VERDICT r04 "next" 5: hashes held by thousands of references (conserved rRNA k-mers) put more pairs into ONE bucket
    of the distribution than it holds.  Until round 4 that sent the whole database to rocPRIM's radix sort and `yacht train`
    off its fused path; now only those buckets go to a side list that is grouped on its own -- yh_db_info says so
    (sort_path = pieces, n_spilled_buckets / n_spilled_pairs) -- and everything equals the oracle: the kept pairs and the
    three statistics of the whole database, the shared-hash count of every reference, and the counts of rows whose pairs
    come from the hot k-mers alone (a low threshold) against the full handle's generic path."""
    import torch

    from yacht_amd import synth
    from yacht_amd.engine import YH_DB_PAIRWISE_ONLY

    refs, hot, holders = _hot_database(77, 12_000, 400, 30, 5_000, 11_000)
    values, offsets = synth.pack(refs)
    n = len(refs)
    c = 0.95 ** 31
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c, threads=oracle.hardware_threads())
    assert wi.size > 500
    with RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY) as tdb, RefDB(values, offsets) as db:
        info = tdb.info()
        assert info["sort_path"] == 3, info  # YH_SORT_PIECES: not the radix fallback
        assert info["n_spilled_buckets"] >= len(hot) * 0.9 and info["n_spilled_buckets"] <= 2 * len(hot) + 2, info
        assert info["n_spilled_pairs"] >= sum(holders) and info["n_spilled_pairs"] <= sum(holders) + 2 * 4096 * info["n_spilled_buckets"], info
        gi, gj, gc = tdb.pairwise(c)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
        assert tdb.index_stats() == wstats
        ns_t = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        ns_d = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        tdb.nshared_device(ns_t.data_ptr())
        tdb.synchronize()
        db.nshared_device(ns_d.data_ptr())
        db.synchronize()
        assert torch.equal(ns_t, ns_d)
        # rows whose pairs are made by the hot k-mers (thousands of columns each): the fused records' list form through the second
        # list array, against the generic handle
        rows = [int(r) for r in np.flatnonzero(np.asarray([np.isin(hot, r).sum() for r in refs[2000:2400]]) >= 3)[:6] + 2000]
        assert rows
        for r0 in rows:
            a = tdb.pairwise(0.004, r0, r0 + 1)
            b = db.pairwise(0.004, r0, r0 + 1)
            assert a[0].size > 1000 and all(np.array_equal(x, y) for x, y in zip(a, b)), r0
    # the same database with the side list compiled out of the decision (YH_NO_SPILL=1): the old behaviour -- refused, radix sort
    env = dict(os.environ, YH_DEBUG_TUNING="1", YH_NO_SPILL="1")
    code = ("import sys, json, numpy as np\nsys.path.insert(0, %r)\nsys.path.insert(0, %r)\n"
            "from test_gpu_keys import _hot_database\nfrom yacht_amd import synth\nfrom yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY\n"
            "refs, hot, holders = _hot_database(77, 12000, 400, 30, 5000, 11000)\nv, o = synth.pack(refs)\n"
            "with RefDB(v, o, flags=YH_DB_PAIRWISE_ONLY) as t:\n    print(json.dumps(t.info()))\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    import json

    old = json.loads(r.stdout.strip().splitlines()[-1])
    assert old["sort_path"] == 1 and old["n_spilled_pairs"] == 0, old  # YH_SORT_RADIX
