"""Batched multi-sample `run`: yh_run_batch must give, for every sample of a
batch, exactly what the one-sample path and the CPU oracle give for that sample alone."""
import numpy as np
import pytest

from oracle import oracle
from yacht_amd import _lib, synth
from yacht_amd.engine import RefDB

pytestmark = pytest.mark.gpu

FULL = 4  # YH_DB_FULL_INDEX


def _samples(values, offsets, n_samples, seed, noise=400):
    rng = np.random.default_rng(seed)
    n = offsets.size - 1
    out = []
    for s in range(n_samples):
        if s == 3:
            out.append(np.zeros(0, dtype=np.uint64))  # an empty sample inside the batch
            continue
        parts = [rng.integers(1, 2**63, size=noise, dtype=np.uint64)]
        for j in rng.choice(n, size=min(n, 1 + s % 7), replace=False):
            r = values[int(offsets[j]):int(offsets[j + 1])]
            if r.size:
                parts.append(r[rng.random(r.size) < rng.uniform(0.05, 0.9)])
        out.append(np.unique(np.concatenate(parts)))
    return out


def _check(values, offsets, samples):
    n = offsets.size - 1
    with RefDB(values, offsets, flags=FULL) as db:
        ov, e, m = db.run_batch(samples)
        assert ov.shape == e.shape == m.shape == (len(samples), n)
        for s, smp in enumerate(samples):
            w_ov = oracle.overlap(values, offsets, smp)
            mask = (w_ov > 0).astype(np.uint8)
            w_e, w_m = oracle.exclusive(values, offsets, mask, smp)
            assert np.array_equal(ov[s], w_ov), f"overlap differs for sample {s}"
            assert np.array_equal(e[s], np.where(mask, w_e, 0)), f"n_excl differs for sample {s}"
            assert np.array_equal(m[s], np.where(mask, w_m, 0)), f"n_match differs for sample {s}"
        # and against the library's own one-sample path
        g_ov, g_e, g_m = db.run_counts(samples[0])
        assert np.array_equal(g_ov, ov[0]) and np.array_equal(g_e, e[0]) and np.array_equal(g_m, m[0])


def test_batch_matches_oracle_clustered(hip_lib):
    values, offsets = synth.config4(seed=31, n_clusters=60, size=700)  # related references: many shared hashes
    _check(values, offsets, _samples(values, offsets, 64, seed=5))


def test_batch_matches_oracle_ragged(hip_lib):
    values, offsets, _ = synth.config2(seed=32)
    for b in (1, 2, 33):
        _check(values, offsets, _samples(values, offsets, b, seed=100 + b))


def test_batch_same_sample_repeated(hip_lib):
    """All 64 bit lanes carry the same sample: every row must be identical to the single run."""
    values, offsets = synth.config4(seed=33, n_clusters=20, size=300)
    smp = _samples(values, offsets, 1, seed=9)[0]
    with RefDB(values, offsets, flags=FULL) as db:
        ov, e, m = db.run_batch([smp] * 64)
        g = db.run_counts(smp)
    for s in range(64):
        assert np.array_equal(ov[s], g[0]) and np.array_equal(e[s], g[1]) and np.array_equal(m[s], g[2])


def test_batch_errors(hip_lib):
    values, offsets = synth.config4(seed=34, n_clusters=4, size=50)
    smp = np.unique(values)[:20]
    with RefDB(values, offsets, flags=16) as db:  # YH_DB_NO_DIRECTORY
        with pytest.raises(_lib.YachtHipError):
            db.run_batch([smp])
    with RefDB(values, offsets, flags=FULL) as db:
        with pytest.raises(_lib.YachtHipError):
            db.run_batch([smp] * 65)
        with pytest.raises(_lib.YachtHipError):
            db.run_batch([smp[::-1].copy()])
