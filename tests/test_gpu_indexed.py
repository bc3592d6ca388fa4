"""GPU: the sample-driven (directory) overlap path, the default handle — same results as the
streaming kernel and the oracle."""
import numpy as np
import pytest

from oracle import oracle
from yacht_amd import synth
from yacht_amd import _lib
from yacht_amd.engine import RefDB, YH_DB_DEFAULT, YH_DB_KEEP_CSR, YH_DB_NO_DIRECTORY

pytestmark = pytest.mark.gpu


def _run_indexed(db, sample):
    import torch

    s = torch.from_numpy(np.ascontiguousarray(sample).view(np.int64).copy()).cuda()
    n = db.n_refs
    ov = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    e = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    m = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    db.run_indexed_device(s.data_ptr(), s.numel(), ov.data_ptr(), e.data_ptr(), m.data_ptr())
    db.synchronize()
    ov2 = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    db.overlap_indexed_device(s.data_ptr(), s.numel(), ov2.data_ptr())
    db.synchronize()
    assert torch.equal(ov, ov2)
    return tuple(t.cpu().numpy().view(np.uint32)[:n] for t in (ov, e, m))


def _check(refs, sample):
    values, offsets = synth.pack(refs)
    want_ov = oracle.overlap(values, offsets, sample)
    want_e, want_m = oracle.exclusive(values, offsets, want_ov > 0, sample)
    with RefDB(values, offsets, flags=YH_DB_DEFAULT) as db:
        ov, e, m = _run_indexed(db, sample)
        assert np.array_equal(ov, want_ov)
        assert np.array_equal(e, want_e) and np.array_equal(m, want_m)
        # the streaming path of the same handle still agrees
        db.set_lookup(_lib.YH_LOOKUP_STREAM)
        assert db.lookup_choice(max(sample.size, 1)) == _lib.YH_LOOKUP_STREAM
        sov, se, sm = db.run_counts(sample)
        assert np.array_equal(sov, want_ov) and np.array_equal(se, want_e) and np.array_equal(sm, want_m)


def test_indexed_small_cases(hip_lib):
    rng = np.random.default_rng(5)
    refs = synth.clustered_refs(rng, 30, (1.0, 0.9, 0.5, 0.25, 0.1), 500)
    refs += [np.zeros(0, np.uint64), np.array([7], np.uint64), refs[0].copy()]
    _check(refs, synth.sample_from_refs(rng, refs, [0, 2, 77, 100, 152], 0.5, 30000))
    # every hash hits; tiny values; hashes at both ends of the range; sample beyond the database
    small = np.unique(rng.integers(0, 3000, 500, dtype=np.uint64))
    big = np.unique(rng.integers(2 ** 63, 2 ** 64 - 1, 2000, dtype=np.uint64, endpoint=True))
    refs2 = [small, small[::2].copy(), big[::3].copy(), np.array([0, 2 ** 64 - 1], np.uint64)]
    _check(refs2, np.unique(np.concatenate([small, big])))
    _check([small, small[::3].copy()], np.unique(np.concatenate([small[::2], big[:100]])))
    _check(refs2, np.zeros(0, np.uint64))
    _check([np.zeros(0, np.uint64)], np.array([1, 2], np.uint64))


def test_indexed_needs_flag(hip_lib):
    from yacht_amd._lib import YH_ERR_UNSUPPORTED, YachtHipError

    with RefDB(np.array([1, 2, 3], np.uint64), np.array([0, 3], np.uint64), flags=YH_DB_NO_DIRECTORY) as db:
        with pytest.raises(YachtHipError) as ei:
            db.overlap_indexed_device(0, 0, 1)
        assert ei.value.code in (YH_ERR_UNSUPPORTED, -1)
        with pytest.raises(YachtHipError):
            db.set_lookup(_lib.YH_LOOKUP_INDEXED)
        assert db.lookup_choice(10) == _lib.YH_LOOKUP_STREAM


def test_indexed_equals_streaming_at_full_scale(hip_lib):
    import torch

    values, offsets, sample = synth.config3_device(seed=77, n_refs=85_205, n_sample=300_000, device="cuda:0")
    n = offsets.numel() - 1
    db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n, flags=YH_DB_DEFAULT | YH_DB_KEEP_CSR)
    try:
        outs = [torch.zeros((3, n), dtype=torch.int32, device="cuda:0") for _ in range(2)]
        torch.cuda.synchronize()
        assert db.lookup_choice(sample.numel()) == _lib.YH_LOOKUP_INDEXED  # 3e5 hashes against 3.3e8: sample-driven by default
        # ... and still for a sample a tenth of the database (measured: 308 us against 533, scripts/probes/crossover.py)
        assert db.lookup_choice(32_000_000) == _lib.YH_LOOKUP_INDEXED
        db.set_lookup(_lib.YH_LOOKUP_STREAM)
        db.run_device(sample.data_ptr(), sample.numel(), outs[0][0].data_ptr(), outs[0][1].data_ptr(),
                      outs[0][2].data_ptr())
        db.set_lookup(_lib.YH_LOOKUP_AUTO)
        db.run_indexed_device(sample.data_ptr(), sample.numel(), outs[1][0].data_ptr(), outs[1][1].data_ptr(),
                              outs[1][2].data_ptr())
        db.synchronize()
        assert torch.equal(outs[0], outs[1])
        assert int((outs[0][0] > 0).sum().item()) >= 200
    finally:
        db.close()
