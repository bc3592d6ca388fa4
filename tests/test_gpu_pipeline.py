"""GPU: the pipelined host-buffer calls (yh_run_submit / yh_run_wait) and the deferred ordering verdict of
yh_run: results equal the oracle whatever is in flight beside them; an unsorted sample is refused by
the check queued ON THE DEVICE (nothing is looked up, the call returns YH_ERR_UNSORTED, the handle stays
usable)."""
import numpy as np
import pytest

from oracle import oracle
from yacht_amd import _lib, synth
from yacht_amd.engine import PinnedArray, RefDB

pytestmark = pytest.mark.gpu


def _db(seed=5, n_refs=1500):
    values, offsets, _ = synth.config3_like(seed=seed, n_refs=n_refs, n_sample=1000, n_present=5)
    return values, offsets


def _samples(values, offsets, k, rng):
    refs = [values[int(offsets[j]):int(offsets[j + 1])] for j in range(offsets.size - 1)]
    out = []
    for i in range(k):
        present = rng.choice(len(refs), size=30, replace=False)
        out.append(synth.sample_from_refs(rng, refs, present, 0.4, 20000 + 3000 * i))
    return out


@pytest.mark.parametrize("pinned", [True, False])
def test_submit_wait_rotating_slots(hip_lib, pinned):
    rng = np.random.default_rng(11)
    values, offsets = _db()
    n = offsets.size - 1
    samples = _samples(values, offsets, 7, rng) + [np.zeros(0, np.uint64), np.array([5], np.uint64)]
    want = []
    for s in samples:
        ov = oracle.overlap(values, offsets, s)
        e, m = oracle.exclusive(values, offsets, ov > 0, s)
        want.append((ov, e, m))
    keep = []

    def buf(k, dt):
        if pinned:
            pa = PinnedArray(k, dt)
            keep.append(pa)
            return pa.array
        return np.zeros(k, dt)

    with RefDB(values, offsets) as db:
        depth = _lib.YH_RUN_SLOTS
        ins = [None] * depth
        outs = [[buf(n, np.uint32) for _ in range(3)] for _ in range(depth)]
        order = [int(x) for x in rng.integers(0, len(samples), size=40)]
        for i, si in enumerate(order + [None] * depth):
            slot = i % depth
            if i >= depth:
                db.run_wait(slot)
                w = want[order[i - depth]]
                assert all(np.array_equal(outs[slot][k], w[k]) for k in range(3)), f"call {i - depth}"
            if si is not None:
                s = samples[si]
                h = buf(max(s.size, 1), np.uint64)[: s.size]
                h[:] = s
                ins[slot] = h
                db.run_submit(slot, h, *outs[slot])
        # a slot in flight cannot be resubmitted; an idle one cannot be waited for
        db.run_submit(0, ins[0] if ins[0] is not None else samples[0], *outs[0])
        with pytest.raises(_lib.YachtHipError):
            db.run_submit(0, samples[0], *outs[0])
        db.run_wait(0)
        with pytest.raises(_lib.YachtHipError):
            db.run_wait(0)
    for pa in keep:
        pa.close()


def test_unsorted_sample_is_refused_on_device(hip_lib):
    rng = np.random.default_rng(3)
    values, offsets = _db(seed=9)
    n = offsets.size - 1
    good = _samples(values, offsets, 2, rng)
    bad = good[0].copy()
    bad[[100, 5000]] = bad[[5000, 100]]
    dup = np.concatenate([good[1][:50], good[1][49:]])  # one repeated hash: not STRICTLY ascending
    ov0 = oracle.overlap(values, offsets, good[0])
    e0, m0 = oracle.exclusive(values, offsets, ov0 > 0, good[0])
    with RefDB(values, offsets) as db:
        outs = [[np.full(n, 7, np.uint32) for _ in range(3)] for _ in range(3)]
        db.run_submit(0, good[0], *outs[0])
        db.run_submit(1, bad, *outs[1])
        db.run_submit(2, good[0], *outs[2])
        db.run_wait(0)
        with pytest.raises(_lib.YachtHipError) as ei:
            db.run_wait(1)
        assert ei.value.code == _lib.YH_ERR_UNSORTED
        db.run_wait(2)
        for k, w in enumerate((ov0, e0, m0)):
            assert np.array_equal(outs[0][k], w) and np.array_equal(outs[2][k], w)
            assert not outs[1][k].any(), "a refused sample must leave all-zero counts"
        # the synchronous call: same verdict, handle still fine afterwards
        for s in (bad, dup):
            with pytest.raises(_lib.YachtHipError) as ei:
                db.run_counts(s)
            assert ei.value.code == _lib.YH_ERR_UNSORTED
        ov, e, m = db.run_counts(good[0])
        assert np.array_equal(ov, ov0) and np.array_equal(e, e0) and np.array_equal(m, m0)
        with pytest.raises(_lib.YachtHipError):
            db.overlap(bad)
        assert np.array_equal(db.overlap(good[0]), ov0)


def test_own_stream_is_ordered_with_the_default_stream(hip_lib):
    """A caller that fills the output buffers with torch (default stream) and reads them back with torch
    needs no synchronize of its own around the *_device calls on the handle's own stream: a long fill
    queued just before the call must not land on top of the results, and the copy queued just after the
    call must see them (include/yacht_hip.h, yh_db_set_stream)."""
    import torch

    rng = np.random.default_rng(77)
    refs = synth.clustered_refs(rng, 400, (1.0, 0.9, 0.5, 0.25, 0.1), 800)
    values, offsets = synth.pack(refs)
    sample = synth.sample_from_refs(rng, refs, list(range(0, len(refs), 7)), 0.5, 20000)
    want_ov = oracle.overlap(values, offsets, sample)
    want_e, want_m = oracle.exclusive(values, offsets, want_ov > 0, sample)
    n = len(refs)
    s = torch.from_numpy(np.ascontiguousarray(sample).view(np.int64).copy()).cuda()
    big = torch.empty(3, 32 << 20, dtype=torch.int32, device="cuda")  # (a fill of 384 MB takes far longer than the step)
    with RefDB(values, offsets) as db:
        for _ in range(20):
            big.fill_(7)
            db.run_device(s.data_ptr(), s.numel(), big[0].data_ptr(), big[1].data_ptr(), big[2].data_ptr())
            got = big[:, :n].cpu().numpy().view(np.uint32)  # no db.synchronize(): the copy is ordered behind the step
            assert np.array_equal(got[0], want_ov) and np.array_equal(got[1], want_e) and np.array_equal(got[2], want_m)
            assert int(big[0, n]) == 7


def _rows_of(ov, e, m):
    keep = np.flatnonzero(ov)
    return keep.astype(np.uint32), ov[keep], e[keep], m[keep]


@pytest.mark.parametrize("pinned", [True, False])
def test_packed_upload_and_compact_rows(hip_lib, pinned):
    """yh_run_submit_packed / yh_run_submit_rows / yh_run_wait_rows mixed with the dense form over all slots: the rows
    are exactly the references with overlap > 0, ascending, with the oracle's three counts; the packed sample is expanded
    on the device to the hashes it was made from."""
    from yacht_amd.engine import ROW_DTYPE, pack_sample

    rng = np.random.default_rng(21)
    values, offsets = _db(seed=6, n_refs=2500)
    n = offsets.size - 1
    samples = _samples(values, offsets, 5, rng) + [np.zeros(0, np.uint64), np.array([5], np.uint64),
                                                   np.unique(np.concatenate([values[::5], np.array([0, 2**64 - 1], np.uint64)]))]
    want = []
    for s in samples:
        ov = oracle.overlap(values, offsets, s)
        e, m = oracle.exclusive(values, offsets, ov > 0, s)
        want.append((ov, e, m))
    keep = []

    def buf(k, dt):
        if pinned:
            pa = PinnedArray(k, dt)
            keep.append(pa)
            return pa.array
        return np.zeros(k, dt)

    with RefDB(values, offsets) as db:
        depth = _lib.YH_RUN_SLOTS
        rows = [buf(n, ROW_DTYPE) for _ in range(depth)]
        dense = [[buf(n, np.uint32) for _ in range(3)] for _ in range(depth)]
        held = [None] * depth
        order = [(int(rng.integers(0, len(samples))), int(rng.integers(0, 3))) for _ in range(60)]
        for i, job in enumerate(order + [None] * depth):
            slot = i % depth
            if i >= depth:
                si, form = order[i - depth]
                w = want[si]
                if form == 2:
                    db.run_wait(slot)
                    assert all(np.array_equal(dense[slot][k], w[k]) for k in range(3)), f"call {i - depth}"
                else:
                    k = db.run_wait_rows(slot)
                    ref, ov, e, m = _rows_of(*w)
                    got = rows[slot][:k]
                    assert k == ref.size and np.array_equal(got["ref"], ref) and np.array_equal(got["overlap"], ov), f"call {i - depth}"
                    assert np.array_equal(got["n_excl"], e) and np.array_equal(got["n_match"], m), f"call {i - depth}"
            if job is not None:
                si, form = job
                s = samples[si]
                if form == 0:    # packed up, rows back
                    p = pack_sample(s)
                    h = buf(max(p.size, 1), np.uint8)[: p.size]
                    h[:] = p
                    held[slot] = h
                    db.run_submit_packed(slot, h, rows[slot])
                elif form == 1:  # raw up, rows back
                    h = buf(max(s.size, 1), np.uint64)[: s.size]
                    h[:] = s
                    held[slot] = h
                    db.run_submit_rows(slot, h, rows[slot])
                else:            # raw up, dense rows back
                    h = buf(max(s.size, 1), np.uint64)[: s.size]
                    h[:] = s
                    held[slot] = h
                    db.run_submit(slot, h, *dense[slot])
        # the synchronous convenience form, both uploads
        for packed in (True, False):
            got = db.run_rows(samples[1], packed=packed)
            ref, ov, e, m = _rows_of(*want[1])
            assert np.array_equal(got["ref"], ref) and np.array_equal(got["overlap"], ov)
            assert np.array_equal(got["n_excl"], e) and np.array_equal(got["n_match"], m)
        # a row buffer that is too small: the number needed comes back with YH_ERR_CAPACITY, the first rows are valid
        small = buf(3, ROW_DTYPE)
        db.run_submit_packed(0, pack_sample(samples[0]), small)
        with pytest.raises(_lib.YachtHipError) as ei:
            db.run_wait_rows(0)
        assert ei.value.code == _lib.YH_ERR_CAPACITY
        ref, ov, e, m = _rows_of(*want[0])
        assert np.array_equal(small["ref"], ref[:3]) and np.array_equal(small["overlap"], ov[:3])
    for pa in keep:
        pa.close()


def test_forged_packed_samples_are_refused(hip_lib):
    """Structure errors are caught at submit (YH_ERR_INVALID_ARG); an ordering the format cannot promise -- blocks that do
    not ascend, a gap that wraps past 2^64 -- by the expansion kernel (YH_ERR_UNSORTED at wait, nothing looked up)."""
    from yacht_amd.engine import ROW_DTYPE, pack_sample

    rng = np.random.default_rng(4)
    values, offsets = _db(seed=9)
    good = _samples(values, offsets, 1, rng)[0]
    p = pack_sample(good)
    n = offsets.size - 1
    with RefDB(values, offsets) as db:
        rows = np.zeros(n, dtype=ROW_DTYPE)
        for bad in (p[:-8], p[:16]):
            with pytest.raises(_lib.YachtHipError) as ei:
                db.run_submit_packed(0, np.ascontiguousarray(bad), rows)
            assert ei.value.code == _lib.YH_ERR_INVALID_ARG
        forged = p.copy()
        forged[32 + 16: 32 + 24] = 0          # first hash of block 1 = 0: not above block 0's last
        db.run_submit_packed(0, forged, rows)
        with pytest.raises(_lib.YachtHipError) as ei:
            db.run_wait_rows(0)
        assert ei.value.code == _lib.YH_ERR_UNSORTED
        wrap = pack_sample(np.array([2**64 - 10, 2**64 - 5, 2**64 - 1], dtype=np.uint64)).copy()
        wrap[32: 32 + 8] = np.frombuffer(np.uint64(2**64 - 3).tobytes(), dtype=np.uint8)   # base so high that the gaps wrap
        db.run_submit_packed(1, wrap, rows)
        with pytest.raises(_lib.YachtHipError) as ei:
            db.run_wait_rows(1)
        assert ei.value.code == _lib.YH_ERR_UNSORTED
        got = db.run_rows(good)               # the handle is fine afterwards
        ov = oracle.overlap(values, offsets, good)
        assert np.array_equal(got["ref"], np.flatnonzero(ov)) and np.array_equal(got["overlap"], ov[ov > 0])


def test_rows_of_a_device_resident_step(hip_lib):
    import torch

    from yacht_amd.engine import ROW_DTYPE

    rng = np.random.default_rng(8)
    values, offsets = _db(seed=12, n_refs=3000)
    n = offsets.size - 1
    s_h = _samples(values, offsets, 1, rng)[0]
    ov = oracle.overlap(values, offsets, s_h)
    e, m = oracle.exclusive(values, offsets, ov > 0, s_h)
    s = torch.from_numpy(s_h.view(np.int64).copy()).cuda()
    c = torch.zeros(3, n, dtype=torch.int32, device="cuda")
    rows = torch.zeros(n, 4, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    with RefDB(values, offsets) as db:
        db.run_device(s.data_ptr(), s.numel(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())
        db.run_rows_device(c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), rows.data_ptr(), n, cnt.data_ptr())
        db.synchronize()
        k = int(cnt.item())
        got = rows[:k].cpu().numpy().view(np.uint32)
        ref = np.flatnonzero(ov)
        assert k == ref.size and np.array_equal(got[:, 0], ref) and np.array_equal(got[:, 1], ov[ref])
        assert np.array_equal(got[:, 2], e[ref]) and np.array_equal(got[:, 3], m[ref])


def test_pipelined_device_steps(hip_lib):
    """yh_run_device_pipelined: the tail of a step runs on a second stream beside the next step's lookup, over two
    alternating sets of counters and step contexts.  Rotating samples of different hit shapes, two and three output
    buffers, other queries in between (which join first), small samples that take other lookup forms."""
    import torch

    rng = np.random.default_rng(31)
    values, offsets = _db(seed=14, n_refs=4000)
    n = offsets.size - 1
    hs = _samples(values, offsets, 5, rng) + [np.zeros(0, np.uint64), np.array([7], np.uint64), values[::2].copy()]
    hs[-1] = np.unique(hs[-1])
    want = []
    for s in hs:
        ov = oracle.overlap(values, offsets, s)
        e, m = oracle.exclusive(values, offsets, ov > 0, s)
        want.append(np.stack([ov, e, m]))
    ds = [torch.from_numpy(s.view(np.int64).copy()).cuda() for s in hs]
    with RefDB(values, offsets) as db:
        for lookup in (_lib.YH_LOOKUP_AUTO, _lib.YH_LOOKUP_INDEXED, _lib.YH_LOOKUP_STREAM):
            db.set_lookup(lookup)
            for nbuf in (2, 3):
                bufs = [torch.zeros(3, n, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
                order = [int(x) for x in rng.integers(0, len(hs), size=24)]
                for i, si in enumerate(order):
                    b = bufs[i % nbuf]
                    db.run_device_pipelined(ds[si].data_ptr(), ds[si].numel(), b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
                    if i >= nbuf - 1 and i % 7 == 3:   # read an older buffer back in the middle of the pipeline
                        db.run_device_join()
                        db.synchronize()
                        j = i - (nbuf - 1)
                        assert np.array_equal(bufs[j % nbuf].cpu().numpy().view(np.uint32), want[order[j]]), (lookup, nbuf, j)
                    if i % 11 == 5:                    # another query in between: joins by itself
                        assert np.array_equal(db.overlap(hs[1]), want[1][0])
                db.synchronize()
                for d in range(nbuf):
                    j = len(order) - 1 - d
                    assert np.array_equal(bufs[j % nbuf].cpu().numpy().view(np.uint32), want[order[j]]), (lookup, nbuf, j)
        db.set_lookup(_lib.YH_LOOKUP_AUTO)
        ov, e, m = db.run_counts(hs[0])              # the plain forms still answer afterwards
        assert np.array_equal(np.stack([ov, e, m]), want[0])


def test_pipelined_device_steps_fused_launches(hip_lib):
    """yh_run_device_pipelined: ONE launch per call that looks up sample k, reduces sample k - 1 and runs the exclusive pass
    of sample k - 2 (k_step_fused in its three geometries: 256-lane workgroups for small samples, 1024-lane tiles of one
    and of two hashes per lane), over two counter sets and three step contexts.  Outputs are complete after two further
    calls or after a join; other queries in between drain the pipeline by themselves."""
    import torch

    rng = np.random.default_rng(41)
    values, offsets = _db(seed=15, n_refs=4000)
    n = offsets.size - 1
    refs = [values[int(offsets[j]):int(offsets[j + 1])] for j in range(n)]
    mh = synth.max_hash_for_scaled(1000)
    hs = []
    for i, size in enumerate((300_000, 600_000, 280_000, 700_000, 540_000)):
        present = rng.choice(n, size=40 + 10 * i, replace=False)
        hs.append(synth.sample_from_refs(rng, refs, present, 0.5, size))
    hs.append(synth.sample_from_refs(rng, refs, rng.choice(n, size=10, replace=False), 0.5, 30_000))  # small: the 256-lane geometry
    hs.append(np.unique(rng.integers(0, mh, size=400_000, dtype=np.uint64)))                         # large, overlaps (almost) nothing
    want = []
    for s in hs:
        ov = oracle.overlap(values, offsets, s, threads=4)
        e, m = oracle.exclusive(values, offsets, ov > 0, s)
        want.append(np.stack([ov, e, m]))
    ds = [torch.from_numpy(s.view(np.int64).copy()).cuda() for s in hs]
    with RefDB(values, offsets) as db:
        assert db.lookup_choice(hs[0].size) == _lib.YH_LOOKUP_INDEXED
        nbuf = 3  # a call's rows are touched by the two launches behind it: three buffers rotate
        bufs = [torch.zeros(3, n, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
        order = [int(x) for x in rng.integers(0, len(hs), size=60)]
        for i, si in enumerate(order):
            b = bufs[i % nbuf]
            db.run_device_pipelined(ds[si].data_ptr(), ds[si].numel(), b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
            if i >= 2 and i % 9 == 4:   # complete after two further calls: read step i - 2 without a join
                torch.cuda.synchronize()
                j = i - 2
                assert np.array_equal(bufs[j % nbuf].cpu().numpy().view(np.uint32), want[order[j]]), ("two calls later", j)
            if i % 13 == 6:             # another query in between: drains the pipeline by itself
                assert np.array_equal(db.overlap(hs[5]), want[5][0])
                for d in range(min(i + 1, nbuf)):
                    j = i - d
                    assert np.array_equal(bufs[j % nbuf].cpu().numpy().view(np.uint32), want[order[j]]), ("after another query", j)
        db.run_device_join()
        db.synchronize()
        for d in range(nbuf):
            j = len(order) - 1 - d
            assert np.array_equal(bufs[j % nbuf].cpu().numpy().view(np.uint32), want[order[j]]), ("after join", j)
        ov, e, m = db.run_counts(hs[1])
        assert np.array_equal(np.stack([ov, e, m]), want[1])


def test_submit_forms_on_a_database_without_hashes(hip_lib):
    """Three empty sketches: every count is zero through every submit form, as through yh_run (found by the fuzzer)."""
    from yacht_amd.engine import ROW_DTYPE, pack_sample

    offsets = np.zeros(4, dtype=np.uint64)
    values = np.zeros(0, dtype=np.uint64)
    sample = np.array([3, 9, 27], dtype=np.uint64)
    with RefDB(values, offsets) as db:
        ov, e, m = db.run_counts(sample)
        assert not ov.any() and not e.any() and not m.any()
        assert db.run_rows(sample).size == 0 and db.run_rows(sample, packed=False).size == 0
        outs = [np.full(3, 7, np.uint32) for _ in range(3)]
        db.run_submit(1, sample, *outs)
        db.run_wait(1)
        assert not any(o.any() for o in outs)
        rows = np.zeros(3, dtype=ROW_DTYPE)
        db.run_submit_packed(2, pack_sample(np.array([5, 4], dtype=np.uint64)[::-1].copy()), rows)
        assert db.run_wait_rows(2) == 0
