"""yh_hyp_test (host C++ of the C ABI, no scipy) against the REFERENCE's own outputs: the parameter grid and the
~16 000 real (n_exclusive, n_matches) pairs that tests/golden/make_golden.py ran through the reference's
single_hyp_test (hypothesis_recovery_src.py:233-306).  Needs no GPU: the library loads and this entry is host code."""
import json
import os

import numpy as np
import pytest

from yacht_amd.hypothesis_recovery_src import hyp_test_batch, hyp_test_native

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL = 1e-12  # relative tolerance of the floating columns; values below TINY (scipy underflows there: it returns 0) compare absolutely
TINY = 1e-250


def close(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    tiny = (np.abs(got) < TINY) & (np.abs(want) < TINY)
    return tiny | (np.abs(got - want) <= REL * np.abs(want))


def test_reference_grid():
    with open(os.path.join(GOLD, "golden_hyp.json")) as f:
        grid = json.load(f)["single_hyp_test"]
    by = {}
    for r in grid:
        by.setdefault((r["k"], r["sig"], r["ani"], r["cov"]), []).append(r)
    n = 0
    for (k, sig, ani, cov), rows in by.items():
        e = [r["e"] for r in rows]
        m = [r["m"] for r in rows]
        got = hyp_test_native(e, m, k, sig, ani, cov)
        want = list(zip(*[r["out"] for r in rows]))
        assert got[0].tolist() == list(want[0]), (k, sig, ani, cov)          # in_sample_est
        assert got[3].tolist() == list(want[3])                              # n_exclusive at coverage
        assert got[5].tolist() == list(want[5])                              # acceptance threshold (integer-valued)
        for col in (1, 6, 7):                                                # p_val, confidence, alt. mutation rate
            ok = close(got[col], want[col])
            assert ok.all(), (col, k, sig, ani, cov, np.asarray(e)[~ok][:3], got[col][~ok][:3], np.asarray(want[col])[~ok][:3])
        n += len(rows)
    assert n == len(grid) >= 800


def test_real_run_tuples():
    z = np.load(os.path.join(GOLD, "golden_hyp_real.npz"))
    with open(os.path.join(GOLD, "golden_hyp_real.json")) as f:
        meta = json.load(f)
    assert z["e"].size == meta["n"] >= 10_000
    for fi, par in enumerate(meta["files"]):
        for cov in (1.0, 0.1, 0.01):
            sel = (z["file_index"] == fi) & (z["cov"] == cov)
            assert sel.sum() > 1000
            got = hyp_test_native(z["e"][sel], z["m"][sel], par["ksize"], par["significance"], par["ani_thresh"], cov)
            assert np.array_equal(got[0], z["present"][sel])
            assert np.array_equal(got[3], z["n_cov"][sel])
            assert np.array_equal(got[5], z["thr"][sel])
            for col, name in ((1, "p_val"), (6, "conf"), (7, "alt")):
                ok = close(got[col], z[name][sel])
                assert ok.all(), (name, par, cov, z["e"][sel][~ok][:3], got[col][~ok][:3], z[name][sel][~ok][:3])


def test_equals_scipy_path_on_random_tuples():
    rng = np.random.default_rng(7)
    for (k, sig, ani, cov) in ((31, 0.99, 0.95, 1), (31, 0.99, 0.95, 0.05), (21, 0.95, 0.9, 0.5), (51, 0.9, 0.99, 0.001)):
        e = np.concatenate([rng.integers(0, 60000, 1500), np.arange(0, 40)])
        m = np.minimum(rng.integers(0, 3000, e.size), e)
        got = hyp_test_native(e, m, k, sig, ani, cov)
        want = hyp_test_batch(e, m, k, sig, ani, cov)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[3], want[3]) and np.array_equal(got[5], want[5])
        for col in (1, 6, 7):
            assert close(got[col], want[col]).all()


def test_edges_and_errors():
    from yacht_amd import _lib

    got = hyp_test_native([0, 0, 1, 5], [0, 1, 1, 9], 31, 0.99, 0.95, 1)
    # e = 0: threshold 0, confidence 0, alt -1 (betaincinv(0, 1, .) is NaN); m > n: p_val 1
    assert got[5].tolist()[:2] == [0.0, 0.0] and got[6][0] == 0.0 and got[7][0] == -1.0
    assert got[0].tolist() == [False, True, True, True] and got[1][3] == 1.0
    assert hyp_test_native([], [], 31)[0].size == 0
    with pytest.raises(_lib.YachtHipError):
        hyp_test_native([1], [1], 0)
    with pytest.raises(_lib.YachtHipError):
        hyp_test_native([1], [1], 31, 1.5)
