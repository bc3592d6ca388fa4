"""The tile kernel streams 32-bit keys (hash >> kshift) and confirms key matches against the 64-bit
hashes afterwards.  These cases are built to make keys collide: sample and database hashes that
differ only in the bits a key drops, runs of equal keys that straddle a tile boundary, and the same
inputs through the 64-bit kernel (YH_WIDE_KEYS=1) as a cross-check.  Everything against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle
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


def _check(values, offsets, sample, hint):
    with RefDB(values, offsets, partitions_hint=hint) as db:
        info = db.info()
        ov, e, m = db.run_counts(sample)
    w_ov = oracle.overlap(values, offsets, sample)
    mask = (w_ov > 0).astype(np.uint8)
    w_e, w_m = oracle.exclusive(values, offsets, mask, sample)
    assert np.array_equal(ov, w_ov)
    assert np.array_equal(e, np.where(mask, w_e, 0))
    assert np.array_equal(m, np.where(mask, w_m, 0))
    return info


def test_key_collisions_are_not_hits(hip_lib):
    values, offsets, sample = _colliding_case()
    info = _check(values, offsets, sample, hint=512)
    assert info["n_partitions"] >= 256  # the case really runs with kshift > 0


def test_key_collisions_other_partitionings(hip_lib):
    values, offsets, sample = _colliding_case(seed=8)
    for hint in (1, 16, 4096):
        _check(values, offsets, sample, hint=hint)


def test_full_range_hashes_single_partition(hip_lib):
    """scaled = 1 sketches: hashes up to 2**64 - 1, few partitions, keys drop up to 31 bits."""
    rng = np.random.default_rng(3)
    refs = [np.unique(rng.integers(0, 2**64 - 1, size=800, dtype=np.uint64, endpoint=True)) for _ in range(40)]
    refs.append(np.array([0, 1, 2**32, 2**32 + 1, 2**63, 2**64 - 2, 2**64 - 1], dtype=np.uint64))
    flat = np.concatenate(refs)
    sample = np.unique(np.concatenate([rng.choice(flat, 3000), rng.choice(flat, 3000) ^ np.uint64(1 << 5),
                                       np.array([0, 2**32 + 1, 2**64 - 1], dtype=np.uint64)]))
    values, offsets = pack_csr(refs)
    for hint in (1, 2, 64):
        _check(values, offsets, sample, hint=hint)


def test_wide_key_build_gives_the_same_counts(hip_lib):
    """The 64-bit tile kernel (YH_WIDE_KEYS=1 at creation) on the collision case, in a child process."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from tests.test_gpu_keys import _colliding_case\n"
        "from yacht_amd.engine import RefDB\n"
        "v, o, s = _colliding_case()\n"
        "db = RefDB(v, o, partitions_hint=512); ov, e, m = db.run_counts(s)\n"
        "np.savez(sys.argv[1], ov=ov, e=e, m=m)\n" % ROOT
    )
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "wide.npz")
        env = dict(os.environ, YH_WIDE_KEYS="1")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, cwd=ROOT, timeout=600)
        wide = np.load(out)
        values, offsets, sample = _colliding_case()
        with RefDB(values, offsets, partitions_hint=512) as db:
            ov, e, m = db.run_counts(sample)
        assert np.array_equal(ov, wide["ov"]) and np.array_equal(e, wide["e"]) and np.array_equal(m, wide["m"])
