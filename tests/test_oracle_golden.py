"""The oracle is pinned before it is trusted: every function of oracle/ against golden vectors
produced by the genuine reference (tests/golden/make_golden.py) and against the reference's own
known-answer tests.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import oracle
from yacht_amd import sigio
from yacht_amd.train_core import format_pair_line

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def train_cases():
    return _load("golden_train.json"), np.load(os.path.join(GOLD, "golden_train.npz"))


@pytest.fixture(scope="module")
def excl_cases():
    return _load("golden_exclusive.json"), np.load(os.path.join(GOLD, "golden_exclusive.npz"))


def pair_lines(pi, pj, pc, sizes):
    return [format_pair_line(int(i), int(j), int(c), int(sizes[i]), int(sizes[j])) for i, j, c in zip(pi, pj, pc)]


@pytest.mark.parametrize("impl", ["cpp", "py"])
def test_train_core_against_reference_exe(train_cases, impl):
    """pairs (as the exact text lines the reference wrote), the three index statistics and the
    greedy selection order, for: exact-threshold pair, ties, duplicates, subset, empty sketch,
    one-hash sketch, N < threads, N = 17 equal sizes, N = 64 / 512 clustered."""
    cases, arrays = train_cases
    for c in cases:
        values, offsets = arrays[c["tag"] + "_values"], arrays[c["tag"] + "_offsets"]
        sizes = np.diff(offsets).astype(np.uint32)
        if impl == "cpp":
            pi, pj, pc, stats = oracle.train_pairs(values, offsets, c["c"], threads=c["threads"])
        else:
            if sizes.size > 100:
                continue  # the pure-Python restatement is for small cases
            pi, pj, pc, stats = oracle.train_pairs_py(values, offsets, c["c"])
        assert pair_lines(pi, pj, pc, sizes) == c["pair_lines"], c["tag"]
        assert stats == (c["stats"]["distinct"], c["stats"]["singletons"], c["stats"]["index"]), c["tag"]
        assert oracle.train_select(sizes, pi, pj).tolist() == c["selected"], c["tag"]
        assert int((sizes == 0).sum()) == c["stats"]["empty"]


def _mask_from_names(names, nontrivial):
    return np.array([n in set(nontrivial) for n in names], dtype=bool)


@pytest.mark.parametrize("impl", ["cpp", "py"])
def test_exclusive_against_reference_python(excl_cases, impl):
    """get_exclusive_hashes of the imported reference: duplicate organism names (both rows in
    the subset), a subset of one, a hash held by three references."""
    cases, arrays = excl_cases
    fn = oracle.exclusive if impl == "cpp" else oracle.exclusive_py
    for c in cases:
        values, offsets = arrays[c["tag"] + "_values"], arrays[c["tag"] + "_offsets"]
        sample = arrays[c["tag"] + "_sample"]
        mask = _mask_from_names(c["names"], c["nontrivial"])
        assert np.flatnonzero(mask).tolist() == c["sub_rows"]
        e, m = fn(values, offsets, mask, sample)
        got = [[int(e[j]), int(m[j])] for j in c["sub_rows"]]
        assert got == c["info"], c["tag"]
        assert not e[~mask].any() and not m[~mask].any()


def test_overlap_cpp_equals_python_sets(excl_cases):
    cases, arrays = excl_cases
    for c in cases:
        values, offsets = arrays[c["tag"] + "_values"], arrays[c["tag"] + "_offsets"]
        sample = arrays[c["tag"] + "_sample"]
        assert np.array_equal(oracle.overlap(values, offsets, sample, threads=3),
                              oracle.overlap_py(values, offsets, sample))


def test_single_hyp_test_against_reference_python():
    """1 500+ (e, m, k, significance, ani, coverage) tuples evaluated by the reference's
    single_hyp_test: decisions, integer columns and thresholds equal; floats to 1e-12."""
    g = _load("golden_hyp.json")
    for row in g["single_hyp_test"]:
        r = oracle.single_hyp_test((row["e"], row["m"]), row["k"], row["sig"], row["ani"], row["cov"])
        w = row["out"]
        assert bool(r[0]) == w[0] and int(r[2]) == w[2] and int(r[3]) == w[3] and int(r[4]) == w[4]
        assert float(r[5]) == w[5]
        for a, b in ((r[1], w[1]), (r[6], w[6]), (r[7], w[7])):
            assert a == pytest.approx(b, rel=1e-12, abs=1e-15)


def test_alt_mut_rate_reference_known_answers():
    """tests/test_unit.py:11-20 and tests/test_unittests.py:86-111 of the reference (np.isclose)."""
    from yacht_amd.hypothesis_recovery_src import get_alt_mut_rate

    for nu, thresh, k, sig, want in _load("golden_hyp.json")["alt_mut_rate_reference_tests"]:
        got = get_alt_mut_rate(nu, thresh, k, sig)
        assert got == -1 if want == -1 else np.isclose(got, want)


def test_fixture_known_answer():
    """The reference's end-to-end assertion (tests/test_workflow.py:62-66): exactly one of the 20
    genomes overlaps sample.sig.zip, with 2 matches out of 3741 exclusive hashes, present at
    min_coverage 0.001 with threshold 0."""
    fx = _load("golden_fixture.json")
    refs = sigio.load_file_as_signatures(os.path.join(GOLD, "fixtures", "20_genomes_sketches.zip"), ksize=31)
    by_md5 = {r.md5sum(): r for r in refs}
    refs = [by_md5[m] for m in fx["md5_order"]]
    sample = sigio.load_file_as_signatures(os.path.join(GOLD, "fixtures", "sample.sig.zip"), ksize=31)[0]
    values = np.concatenate([r.minhash.mins for r in refs])
    offsets = np.concatenate([[0], np.cumsum([len(r.minhash) for r in refs])]).astype(np.uint64)
    ov = oracle.overlap(values, offsets, sample.minhash.mins)
    assert ov.tolist() == fx["overlap_python_sets"]
    assert int((ov > 0).sum()) == 1 and int(ov.max()) == 2
    e, m = oracle.exclusive(values, offsets, ov > 0, sample.minhash.mins)
    j = int(np.flatnonzero(ov)[0])
    row = fx["rows"][0]
    assert refs[j].name == row["organism_name"] == "CP032507.1 Ectothiorhodospiraceae bacterium BW-2 chromosome, complete genome"
    assert (int(e[j]), int(m[j])) == (row["n_exclusive"], row["n_matches"]) == (3741, 2)
    r = oracle.single_hyp_test((3741, 2), 31, 0.99, 0.95, 0.001)
    assert bool(r[0]) is True and r[5] == 0.0 and r[3] == 3
    assert r[1] == pytest.approx(row["hyp_cov_0.001"][1], rel=1e-12)


def test_oracle_train_core_at_config3_full_size_equals_the_genuine_reference():
    """The oracle's restatement of main.cpp:215-407 on BASELINE configs[3] at its real size (10 000 sketches, 5e7 hashes)
    against what the genuine reference executable wrote for the same input (tests/golden/golden_train_cfg3.json, made by
    make_golden.py cfg3 in the build container): 20 000 pair lines by digest, the three statistics, the selection order.
    (~1 minute on 8 cores: the reference's own index build is single-threaded.)"""
    import hashlib
    import json

    from yacht_amd import synth
    from yacht_amd.train_core import format_pair_line

    with open(os.path.join(GOLD, "golden_train_cfg3.json")) as f:
        g = json.load(f)
    values, offsets = synth.config4()
    assert hashlib.sha256(values.tobytes() + offsets.tobytes()).hexdigest() == g["input_sha256"], "the generator drifted"
    sizes = np.diff(offsets).astype(np.uint32)
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, g["c"], threads=8)
    lines = [format_pair_line(int(i), int(j), int(k), int(sizes[i]), int(sizes[j])) for i, j, k in zip(wi, wj, wc)]
    assert len(lines) == g["n_pair_lines"]
    assert hashlib.sha256("\n".join(lines).encode()).hexdigest() == g["pair_lines_sha256"]
    assert tuple(wstats) == (g["stats"]["distinct"], g["stats"]["singletons"], g["stats"]["index"])
    assert oracle.train_select(sizes, wi, wj).tolist() == g["selected"]
