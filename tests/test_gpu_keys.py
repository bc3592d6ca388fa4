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
