"""GPU box: (1) the hypothesis-test grids again -- they are host arithmetic, but the scipy of THAT box is
what turns the GPU's counts into `in_sample_est`, so the 1 512-tuple reference grid and the
batch == scalar check also run under `-m gpu` (same functions as the CPU suite, called through);
(2) `yacht run`'s orchestration goes through ONE fused library call (yh_run) and still answers
caller-supplied name lists through the general path; (3) the real hit shape (~29 % of the
references overlap the sample) against the oracle."""
import numpy as np
import pandas as pd
import pytest

from oracle import oracle
from yacht_amd import hypothesis_recovery_src as hr
from yacht_amd import synth
from yacht_amd.engine import RefDB

pytestmark = pytest.mark.gpu


def test_hyp_grid_on_this_box(hip_lib):
    import scipy

    import test_host
    import test_oracle_golden

    print("scipy", scipy.__version__)
    test_host.test_hyp_test_batch_equals_scalar_and_golden()
    test_oracle_golden.test_single_hyp_test_against_reference_python()
    test_oracle_golden.test_alt_mut_rate_reference_known_answers()
    test_host.test_single_hyp_test_return_types()


def test_native_hyp_test_on_this_box(hip_lib):
    """yh_hyp_test -- the long-double arithmetic that decides in_sample_est for a C caller -- on the GPU box's own host
    (libm, CPU): the reference's 1 512-tuple grid and its 16 344 real (n_exclusive, n_matches) pairs."""
    import test_hyp_native

    test_hyp_native.test_reference_grid()
    test_hyp_native.test_real_run_tuples()
    test_hyp_native.test_equals_scipy_path_on_random_tuples()
    test_hyp_native.test_edges_and_errors()


class _Sig:
    class _MH:
        def __init__(self, mins):
            self.mins = mins

    def __init__(self, mins):
        self.minhash = self._MH(mins)


def test_run_path_uses_one_fused_call(hip_lib, monkeypatch):
    rng = np.random.default_rng(4)
    refs = synth.clustered_refs(rng, 30, (1.0, 0.9, 0.5, 0.25, 0.1), 500)
    values, offsets = synth.pack(refs)
    n = len(refs)
    sample = synth.sample_from_refs(rng, refs, [0, 1, 7, 60, 61, 140], 0.5, 30000)
    manifest = pd.DataFrame({"organism_name": [f"org{j}" for j in range(n)], "md5sum": [f"{j:032x}" for j in range(n)]})
    want_ov = oracle.overlap(values, offsets, sample)
    dup = int(np.flatnonzero(want_ov == 0)[0])
    manifest.loc[dup, "organism_name"] = "org0"  # duplicate name: reference `dup` has no overlap but is selected with org0
    calls = {"run": 0, "excl": 0}
    with RefDB(values, offsets) as db:
        monkeypatch.setattr(hr, "get_reference_db", lambda *a, **k: db)
        orig_run, orig_excl = db.run_counts, db.exclusive
        db.run_counts = lambda s: (calls.__setitem__("run", calls["run"] + 1), orig_run(s))[1]
        db.exclusive = lambda mk, s: (calls.__setitem__("excl", calls["excl"] + 1), orig_excl(mk, s))[1]
        # what get_organisms_with_nonzero_overlap does with the handle
        mins = np.ascontiguousarray(sample, dtype=np.uint64)
        ov, e, m = db.run_counts(mins)
        hr._LAST_RUN.clear()
        hr._LAST_RUN.update(db=db, mins=mins, overlap=ov, n_excl=e, n_match=m)
        assert np.array_equal(ov, want_ov)
        # (a) exactly the overlapping organisms, unique names: served from the fused call
        m2 = manifest.copy()
        m2.loc[dup, "organism_name"] = f"org{dup}"
        names = [m2["organism_name"][j] for j in np.flatnonzero(want_ov)]
        info, sub = hr.get_exclusive_hashes(m2, names, _Sig(sample), 31, "unused")
        we, wm = oracle.exclusive(values, offsets, want_ov > 0, sample)
        rows = np.flatnonzero(want_ov)
        assert info == [(int(we[j]), int(wm[j])) for j in rows] and len(sub) == rows.size
        assert calls == {"run": 1, "excl": 0}
        # (b) the duplicate name pulls reference 5 in: a different subset -> the general path, still exact
        names = [manifest["organism_name"][j] for j in np.flatnonzero(want_ov)]
        sel = manifest["organism_name"].isin(names).to_numpy()
        assert sel[dup] and want_ov[dup] == 0 and want_ov[0] > 0
        info, sub = hr.get_exclusive_hashes(manifest, names, _Sig(sample), 31, "unused")
        we, wm = oracle.exclusive(values, offsets, sel, sample)
        assert info == [(int(we[j]), int(wm[j])) for j in np.flatnonzero(sel)]
        assert calls == {"run": 1, "excl": 1}
        hr._LAST_RUN.clear()


@pytest.mark.parametrize("n_refs", [3000, 12000])
def test_real_hit_shape_against_oracle(hip_lib, n_refs):
    """~29 % of the references overlap an (almost all hits) sample: the hit-table overflow of the
    lookup kernel and the dense regime of the exclusive pass."""
    values, offsets, _ = synth.config3_like(seed=21, n_refs=n_refs, n_sample=1000, n_present=3)
    refs = [values[int(offsets[j]):int(offsets[j + 1])] for j in range(n_refs)]
    rng = np.random.default_rng(8)
    sample = synth.real_shape_sample(rng, refs, n_sample=int(0.97 * n_refs))
    want_ov = oracle.overlap(values, offsets, sample, threads=4)
    assert 0.2 * n_refs < int((want_ov > 0).sum()) < 0.4 * n_refs
    we, wm = oracle.exclusive(values, offsets, want_ov > 0, sample)
    with RefDB(values, offsets) as db:
        ov, e, m = db.run_counts(sample)
        assert np.array_equal(ov, want_ov) and np.array_equal(e, we) and np.array_equal(m, wm)
        e2, m2 = db.exclusive(want_ov > 0, sample)  # the general path on the same subset
        assert np.array_equal(e2, we) and np.array_equal(m2, wm)
        half = (want_ov > 0) & (np.arange(n_refs) % 2 == 0)
        we, wm = oracle.exclusive(values, offsets, half, sample)
        e3, m3 = db.exclusive(half, sample)
        assert np.array_equal(e3, we) and np.array_equal(m3, wm)


def test_holder_sets_are_the_distinct_ones(hip_lib):
    """yh_db_info.n_holder_sets: the run step's exclusive pass walks one record per DISTINCT (reference, set of other
    holders of a shared hash) -- counted here with Python sets from the CSR -- and the counts it produces from
    them equal the oracle's (clusters, a hash held by every reference, duplicates of whole sketches)."""
    rng = np.random.default_rng(99)
    refs = synth.clustered_refs(rng, 120, (1.0, 0.9, 0.5, 0.25, 0.1), 600)
    everywhere = np.array([7, 9], np.uint64)  # held by all: holder lists longer than the seven inline slots
    refs = [np.unique(np.concatenate([r, everywhere])) for r in refs] + [refs[3].copy(), np.zeros(0, np.uint64)]
    values, offsets = synth.pack(refs)
    holders = {}
    for j, r in enumerate(refs):
        for h in r.tolist():
            holders.setdefault(h, []).append(j)
    want_sets = 0
    for j, r in enumerate(refs):  # (a list of more than seven others is kept by reference to its postings: one record per hash)
        want_sets += len({(tuple(o for o in holders[h] if o != j) if len(holders[h]) <= 8 else ("long", h))
                          for h in r.tolist() if len(holders[h]) > 1})
    sample = synth.sample_from_refs(rng, refs, list(range(0, len(refs), 5)), 0.6, 5000)
    want_ov = oracle.overlap(values, offsets, sample)
    want_e, want_m = oracle.exclusive(values, offsets, want_ov > 0, sample)
    with RefDB(values, offsets) as db:
        info = db.info()
        assert info["n_holder_sets"] == want_sets and want_sets < info["n_shared_postings"]
        ov, e, m = db.run_counts(sample)
        assert np.array_equal(ov, want_ov) and np.array_equal(e, want_e) and np.array_equal(m, want_m)
