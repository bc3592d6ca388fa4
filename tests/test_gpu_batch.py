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


def _want(values, offsets, smp):
    """The three count rows of one sample by the oracle, zero outside the sample's subset (what the batch calls leave)."""
    w_ov = oracle.overlap(values, offsets, smp)
    mask = (w_ov > 0).astype(np.uint8)
    w_e, w_m = oracle.exclusive(values, offsets, mask, smp)
    return w_ov, np.where(mask, w_e, 0), np.where(mask, w_m, 0)


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


@pytest.mark.parametrize("b", [65, 128, 200, 256])
def test_batch_of_more_than_64_samples(hip_lib, b):
    """Round 6 (ABI 8): up to 256 samples per pass -- the subset words in planes of 64 samples.  Related references (many shared
    hashes: the exclusive pass works plane by plane), an empty sample, ragged lengths; every sample against the oracle."""
    values, offsets = synth.config4(seed=36, n_clusters=30, size=200)
    _check(values, offsets, _samples(values, offsets, b, seed=200 + b, noise=150))


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
            db.run_batch([smp] * 257)
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
        # the block's subset words in compact form (ABI 5): packed -> "gathered" (two ranks: the real words and a second,
        # disjoint-ish set) -> OR; at the default capacity of dist.BatchedRangeRunner (320 per sample) and undersized
        words = torch.from_numpy(np.bitwise_or.reduce((got[0] > 0).astype(np.uint64) << np.arange(64, dtype=np.uint64)[:, None], axis=0).view(np.int64)).to("cuda:0")
        other = torch.roll(words, 12345) & 0x0F0F
        want = (words | other).cpu().numpy()
        nz = int((words != 0).sum().item())
        for cap, expect_ovf in ((320 * 64, 0), (nz - 1, 1), (n_refs, 0)):
            L = int(_lib.load().yh_run_batch_words_packed_len(cap))
            assert L == 1 + cap + (cap + 1) // 2
            gath = torch.full((2, L), -1, dtype=torch.int64, device="cuda:0")
            ored = torch.full((n_refs,), -1, dtype=torch.int64, device="cuda:0")
            ovf = torch.full((1,), 7, dtype=torch.int32, device="cuda:0")
            db.synchronize()
            torch.cuda.synchronize()
            db.run_batch_words_pack_device(words.data_ptr(), gath[0].data_ptr(), cap)
            db.run_batch_words_pack_device(other.data_ptr(), gath[1].data_ptr(), cap)
            db.run_batch_words_unpack_device(gath.data_ptr(), 2, cap, ored.data_ptr(), ovf.data_ptr())
            db.synchronize()
            assert int(gath[0, 0].item()) == nz and int(gath[1, 0].item()) == int((other != 0).sum().item())
            assert int(ovf.item()) == expect_ovf, (cap, nz)
            if not expect_ovf:
                assert np.array_equal(ored.cpu().numpy(), want), f"OR of the packed words differs (cap {cap})"
        assert nz <= 320 * 64, f"{nz} non-zero subset words in a block of 64: the runner's default capacity is too small for the bench workload"
    finally:
        db.close()


def test_batch_words_compact_small(hip_lib):
    """yh_run_batch_words_pack_device / _unpack_device on small rows: empty, full, forged reference ids, one rank."""
    import torch

    values, offsets = synth.config4(seed=35, n_clusters=30, size=60)
    n = offsets.size - 1
    rng = np.random.default_rng(3)
    with RefDB(values, offsets, flags=FULL) as db:
        for density in (0.0, 0.07, 1.0):
            w = rng.integers(1, 2 ** 63, size=n, dtype=np.int64) * (rng.random(n) < density)
            wt = torch.from_numpy(w).to("cuda:0")
            cap = n
            L = 1 + cap + (cap + 1) // 2
            g = torch.zeros((1, L), dtype=torch.int64, device="cuda:0")
            out = torch.full((n,), 5, dtype=torch.int64, device="cuda:0")
            ovf = torch.zeros(1, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()
            db.run_batch_words_pack_device(wt.data_ptr(), g.data_ptr(), cap)
            db.run_batch_words_unpack_device(g.data_ptr(), 1, cap, out.data_ptr(), ovf.data_ptr())
            db.synchronize()
            assert np.array_equal(out.cpu().numpy(), w) and int(ovf.item()) == 0
        # a forged buffer: count beyond the capacity, reference ids beyond N -> flagged / ignored, nothing written out of bounds
        cap = 8
        L = 1 + cap + (cap + 1) // 2
        forged = np.zeros(L, dtype=np.int64)
        forged[0] = 1000
        forged[1:1 + cap] = 3
        forged[1 + cap:].view(np.uint32)[:cap] = [0, n + 5, 2 ** 31, 1, n, 0xFFFFFFFF, 2, 3]
        g = torch.from_numpy(forged).to("cuda:0")
        out = torch.zeros(n, dtype=torch.int64, device="cuda:0")
        ovf = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        db.run_batch_words_unpack_device(g.data_ptr(), 1, cap, out.data_ptr(), ovf.data_ptr())
        db.synchronize()
        o = out.cpu().numpy()
        assert int(ovf.item()) == 1 and o[:4].tolist() == [3, 3, 3, 3] and not o[4:].any()


@pytest.mark.parametrize("finish_stream", [True, False])
def test_runner_second_halves_on_their_own_stream(hip_lib, finish_stream):
    """dist.BatchedRangeRunner over one hash range (no process group): many blocks through the three batch slots with the second
    halves on the finish stream (ABI 6: yh_db_set_batch_finish_stream) or on the handle's one stream -- every block against
    the oracle, an undersized word exchange repeated on the way, and a single-sample query of the same handle afterwards
    (which must come behind the last second half: it uses the same work list and subset bits)."""
    import torch

    from yacht_amd import dist as ydist

    values, offsets = synth.config4(seed=41, n_clusters=60, size=90)
    n = offsets.size - 1
    dev = torch.device("cuda", 0)
    vt = torch.from_numpy(values.view(np.int64).copy()).to(dev)
    ot = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    blocks = [_samples(values, offsets, 1 + (7 * j) % 9, seed=500 + j) for j in range(14)]
    hr = ydist.HashRangeRefDB(vt, ot, [0, 2 ** 64], ydist.HipRangeBackend(0))
    try:
        for kw in (dict(), dict(cap_words=2), dict(dense_rows=True), dict(compact_words=False),
                   # blocks of up to 200 samples: three word planes through the same three slots (round 6)
                   dict(batch=200), dict(batch=200, cap_words=7, cap_rows=11), dict(batch=256, compact_words=False)):
            if "batch" in kw:
                blocks = [_samples(values, offsets, nb, seed=900 + j, noise=60) for j, nb in enumerate((200, 65, 3, 130, 64, 200, 1, 199))]
            kw = dict(dict(batch=9), **kw)
            got = {}

            def on_result(tag, n_in, rows, dense):
                got[tag] = (dense[:, :n_in].clone() if rows is None else ydist.BatchRowsReducer.rows_to_dense(rows, n_in, n)).cpu().numpy()

            run = ydist.BatchedRangeRunner(hr, dst=0, nbuf=3, on_result=on_result, finish_stream=finish_stream, **kw)
            assert (run.s2 is not None) == finish_stream
            packed = [hr.pack_batch([torch.from_numpy(s.view(np.int64).copy()).to(dev) for s in blk]) for blk in blocks]
            for rep in range(2):
                for j, blk in enumerate(blocks):
                    run.submit(packed[j], len(blk), tag=(rep, j))
                run.drain()
            if "cap_words" in kw:
                assert run.n_words_overflow >= 1
            # a single-sample query right behind the last block, no synchronisation in between
            one = torch.from_numpy(blocks[3][0].view(np.int64).copy()).to(dev)
            cnt = torch.zeros((3, n), dtype=torch.int32, device=dev)
            run.submit(packed[5], len(blocks[5]), tag="last")
            run.drain()
            hr.local.handle.run_device(one.data_ptr(), one.numel(), cnt[0].data_ptr(), cnt[1].data_ptr(), cnt[2].data_ptr())
            hr.local.handle.synchronize()
            want = _want(values, offsets, blocks[3][0])
            assert all(np.array_equal(cnt[k].cpu().numpy(), want[k]) for k in range(3)), f"single-sample query behind the runner ({kw})"
            run.close()
            assert sorted(k for k in got if k != "last") == [(rep, j) for rep in range(2) for j in range(len(blocks))]
            for key, dense in got.items():
                blk = blocks[5] if key == "last" else blocks[key[1]]
                for k, smp in enumerate(blk):
                    want = _want(values, offsets, smp)
                    for row in range(3):
                        assert np.array_equal(dense[row, k], want[row]), f"runner {kw} finish_stream={finish_stream} block {key} sample {k} row {row}"
    finally:
        hr.close()
