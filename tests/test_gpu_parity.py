"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Every test here calls libyacht_hip.so via yacht_amd.engine (ctypes) and compares uint32 counts
and index lists with oracle/ on the same seeded inputs.
"""
import numpy as np
import pytest

from oracle import oracle
from yacht_amd import _lib, synth
from yacht_amd.engine import RefDB, YH_DB_KEEP_CSR, train_select

pytestmark = pytest.mark.gpu

C_DEFAULT = 0.95 ** 31


def _check_all(refs, sample, c_thresh=C_DEFAULT):
    values, offsets = synth.pack(refs)
    sizes = np.diff(offsets).astype(np.uint32)
    with RefDB(values, offsets, flags=YH_DB_KEEP_CSR) as db:
        want = oracle.overlap(values, offsets, sample)
        # every query through the streaming kernel, through the sample-driven one, and by the library's own choice
        for mode in (_lib.YH_LOOKUP_STREAM, _lib.YH_LOOKUP_INDEXED, _lib.YH_LOOKUP_AUTO):
            db.set_lookup(mode)
            # R1
            got = db.overlap(sample)
            assert np.array_equal(got, want)
            assert np.array_equal(db.overlap(sample, method="bsearch"), want)
            # R2 on the overlap>0 subset and on two arbitrary subsets
            rng = np.random.default_rng(7)
            masks = [want > 0, np.ones(len(refs), bool), rng.random(len(refs)) < 0.5]
            for mask in masks:
                we, wm = oracle.exclusive(values, offsets, mask, sample)
                ge, gm = db.exclusive(mask, sample)
                assert np.array_equal(ge, we)
                assert np.array_equal(gm, wm)
            ov, e, m = db.run_counts(sample)
            we, wm = oracle.exclusive(values, offsets, want > 0, sample)
            assert np.array_equal(ov, want) and np.array_equal(e, we) and np.array_equal(m, wm)
        # T2-T5
        wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c_thresh, threads=3)
        gi, gj, gc = db.pairwise(c_thresh)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
        assert db.index_stats() == wstats
        assert np.array_equal(train_select(sizes, gi, gj), oracle.train_select(sizes, wi, wj))
        return db.info()


def test_clustered_small(hip_lib):
    rng = np.random.default_rng(1)
    refs = synth.clustered_refs(rng, 40, (1.0, 0.9, 0.5, 0.25, 0.1), 400)
    refs.append(np.zeros(0, np.uint64))          # empty sketch
    refs.append(np.array([12345], np.uint64))    # one-hash sketch
    refs.append(refs[0].copy())                  # exact duplicate
    refs.append(refs[5][::2].copy())             # strict subset
    sample = synth.sample_from_refs(rng, refs, [0, 3, 7, 12, 100, 150], 0.5, 20000)
    info = _check_all(refs, sample)
    assert info["n_refs"] == len(refs)


def test_micro_golden_layout(hip_lib):
    """The worked example of SURVEY.md §8c (tiny hash values, exact-threshold pair, ties)."""
    A = np.arange(1, 1001, dtype=np.uint64)
    B = np.concatenate([np.arange(1, 251), np.arange(5000, 6750)]).astype(np.uint64)
    X = np.arange(10000, 10500, dtype=np.uint64)
    Z = np.arange(10000, 10100, dtype=np.uint64)
    refs = [A, B, X, X.copy(), Z, np.zeros(0, np.uint64), np.array([20000], np.uint64)]
    sample = np.unique(np.concatenate([A[::3], X[::7], np.array([20000, 999999], np.uint64)]))
    _check_all(refs, sample, c_thresh=0.25)
    values, offsets = synth.pack(refs)
    with RefDB(values, offsets) as db:
        gi, gj, gc = db.pairwise(0.25)
        assert list(zip(gi.tolist(), gj.tolist(), gc.tolist())) == [
            (0, 1, 250), (2, 3, 500), (3, 2, 500), (4, 2, 100), (4, 3, 100)]
        sel = train_select(np.diff(offsets).astype(np.uint32), gi, gj)
        assert sel.tolist() == [5, 6, 3, 1]
        assert db.index_stats() == (3251, 2501, 750)


def test_every_hash_hits_and_subtiles(hip_lib):
    """Sample = union of all references (every lookup is a hit): a streaming workgroup's slice of the sample is far
    larger than one LDS tile and the kernel must loop over sub-tiles."""
    rng = np.random.default_rng(3)
    refs = synth.independent_refs(rng, 60, 900, 0.5, 50, 5000)
    sample = np.unique(np.concatenate(refs + [synth.random_sketch(rng, 40000, synth.max_hash_for_scaled(1000))]))
    _check_all(refs, sample)


def test_one_hash_in_hundreds_of_references(hip_lib):
    """200 identical sketches + 50 that share half of it: one probe of the sample matches a run of up
    to 250 equal stream elements -- more than a wave's candidate queue holds -- and the runs cross
    lane, block and super-block boundaries."""
    rng = np.random.default_rng(21)
    core = synth.random_sketch(rng, 300, synth.max_hash_for_scaled(1000))
    refs = [core.copy() for _ in range(200)] + [np.unique(np.concatenate([core[::2], synth.random_sketch(
        rng, 150, synth.max_hash_for_scaled(1000))])) for _ in range(50)]
    sample = np.unique(np.concatenate([core, synth.random_sketch(rng, 5000, synth.max_hash_for_scaled(1000))]))
    _check_all(refs, sample)
    _check_all(refs, core[::3].copy())


def test_hash_extremes(hip_lib):
    """Hashes near 0 and 2**64-1 (scaled=1 sketches), sample hashes outside the database range."""
    rng = np.random.default_rng(4)
    big = np.unique(rng.integers(2 ** 63, 2 ** 64 - 1, size=3000, dtype=np.uint64, endpoint=True))
    small = np.unique(rng.integers(0, 5000, size=600, dtype=np.uint64))
    refs = [big[::2].copy(), big[1::3].copy(), small, np.array([0, 2 ** 64 - 1], np.uint64)]
    sample = np.unique(np.concatenate([big[::5], small[::2], np.array([0, 2 ** 64 - 1], np.uint64)]))
    _check_all(refs, sample)
    # database confined to small values, sample reaching far above the last partition
    refs2 = [small, small[::2].copy(), np.arange(100, 200, dtype=np.uint64)]
    sample2 = np.unique(np.concatenate([small[::3], big[:500]]))
    _check_all(refs2, sample2)


def test_empty_inputs(hip_lib):
    refs = [np.zeros(0, np.uint64), np.zeros(0, np.uint64)]
    _check_all(refs, np.array([1, 2, 3], np.uint64))
    rng = np.random.default_rng(5)
    refs = synth.independent_refs(rng, 5, 300, 0.2, 10, 1000)
    _check_all(refs, np.zeros(0, np.uint64))


def test_unsorted_rejected(hip_lib):
    from yacht_amd._lib import YH_ERR_UNSORTED, YachtHipError

    v = np.array([5, 4, 9], np.uint64)
    o = np.array([0, 3], np.uint64)
    with pytest.raises(YachtHipError) as ei:
        RefDB(v, o)
    assert ei.value.code == YH_ERR_UNSORTED
    with RefDB(np.array([1, 2, 3], np.uint64), o) as db:
        with pytest.raises(YachtHipError):
            db.overlap(np.array([3, 3], np.uint64))


def test_config2_overlap_and_exclusive(hip_lib):
    """BASELINE.json configs[1]: 1 000 refs x ~5 000 hashes vs a 1 M-hash sample."""
    values, offsets, sample = synth.config2()
    want = oracle.overlap(values, offsets, sample, threads=4)
    with RefDB(values, offsets, flags=YH_DB_KEEP_CSR) as db:
        assert np.array_equal(db.overlap(sample), want)
        assert np.array_equal(db.overlap(sample, method="bsearch"), want)
        ov, e, m = db.run_counts(sample)
        we, wm = oracle.exclusive(values, offsets, want > 0, sample)
        assert np.array_equal(e, we) and np.array_equal(m, wm)
        assert int((want > 0).sum()) >= 50


def test_config4_small_train(hip_lib):
    """configs[3] shape at 400 clusters x 5: pairs, stats and the greedy selection."""
    values, offsets = synth.config4(n_clusters=400, size=2000)
    sizes = np.diff(offsets).astype(np.uint32)
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, C_DEFAULT, threads=4)
    with RefDB(values, offsets) as db:
        gi, gj, gc = db.pairwise(C_DEFAULT)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
        assert db.index_stats() == wstats
        # row-block form used by the multi-GPU sharding: concatenation of blocks == whole
        parts = [db.pairwise(C_DEFAULT, a, b) for a, b in ((0, 700), (700, 701), (701, len(sizes)))]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), wi)
        assert np.array_equal(np.concatenate([p[2] for p in parts]), wc)
    assert np.array_equal(train_select(sizes, gi, gj), oracle.train_select(sizes, wi, wj))
