"""Batched multi-sample `run`: yh_run_batch must give, for every sample of a
batch, exactly what the one-sample path and the CPU oracle give for that sample alone."""
import numpy as np
import pytest

from oracle import oracle
from yacht_amd import _lib, synth
from yacht_amd.engine import RefDB

pytestmark = pytest.mark.gpu

FULL = 0  # YH_DB_DEFAULT: the directory is part of every handle unless YH_DB_NO_DIRECTORY


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


def test_batch_rs214_scale_against_oracle(hip_lib):
    """yh_run_batch_device at the bench's own scale -- 85 205 references (3.3e8 hashes), 64 DISTINCT 1e6-hash samples in
    one pass -- against the ORACLE on the whole database for three of them (first, middle, last), and the compact rows
    of the batch against the dense rows for all 64 (the bench checked this only against the single step)."""
    import torch

    from yacht_amd.engine import YH_DB_DEFAULT

    n_refs = 85_205
    plan = synth.global_db_plan(1002, n_refs, cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
    values, offsets = synth.global_db_refs_device(plan, np.arange(n_refs), device="cuda:0")
    samples = [synth.global_db_sample_device(plan, 5000 + i, n_sample=1_000_000, n_present=200, device="cuda:0") for i in range(64)]
    cat = torch.cat(samples).contiguous()
    soff = torch.zeros(65, dtype=torch.int64, device="cuda:0")
    soff[1:] = torch.cumsum(torch.tensor([int(x.numel()) for x in samples], dtype=torch.int64, device="cuda:0"), 0)
    out = torch.zeros((3, 64, n_refs), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_refs, flags=YH_DB_DEFAULT)
    try:
        db.run_batch_device(cat.data_ptr(), soff.data_ptr(), 64, int(cat.numel()), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
        cap = 64 * 2048
        vals = torch.zeros((cap, 3), dtype=torch.int32, device="cuda:0")
        rows = torch.zeros((cap, 5), dtype=torch.int32, device="cuda:0")
        n1 = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        n2 = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        db.run_batch_rows_pack_device(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), vals.data_ptr(), cap, n1.data_ptr())
        db.run_batch_rows_unpack_device(vals.data_ptr(), cap, rows.data_ptr(), n2.data_ptr())
        db.synchronize()
        got = out.cpu().numpy().view(np.uint32)
        hv, ho = values.cpu().numpy().view(np.uint64), offsets.cpu().numpy().astype(np.uint64)
        for s in (0, 31, 63):
            hs = samples[s].cpu().numpy().view(np.uint64)
            w_ov = oracle.overlap(hv, ho, hs, threads=8)
            w_e, w_m = oracle.exclusive(hv, ho, w_ov > 0, hs)
            assert np.array_equal(got[0][s], w_ov), f"overlap differs for sample {s}"
            assert np.array_equal(got[1][s], w_e), f"n_excl differs for sample {s}"
            assert np.array_equal(got[2][s], w_m), f"n_match differs for sample {s}"
        k = int(n1.item())
        assert k == int(n2.item()) == int((got[0] > 0).sum()) and k <= cap
        r = rows[:k].cpu().numpy().view(np.uint32)
        dense = np.zeros((3, 64, n_refs), dtype=np.uint32)
        for c in range(3):
            dense[c, r[:, 0], r[:, 1]] = r[:, 2 + c]
        assert np.array_equal(dense, got), "the compact rows of the batch do not reproduce its dense rows"
        assert np.array_equal(np.lexsort((r[:, 0], r[:, 1])), np.arange(k)), "(reference, sample) order"
    finally:
        db.close()
