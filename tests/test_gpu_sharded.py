"""GPU: the posting-shard entry points (yh_db_create_from_pairs, yh_exclusive_partial_device,
yh_db_nshared_device, yh_exclusive_finalize_device) and ShardedRun on the HIP backend."""
import os

import numpy as np
import pytest

from oracle import oracle
from yacht_amd import dist as ydist
from yacht_amd import synth
from yacht_amd.engine import RefDB, YH_DB_NO_INDEX

pytestmark = pytest.mark.gpu

SIGN = -(2 ** 63)


def _data():
    rng = np.random.default_rng(99)
    refs = synth.clustered_refs(rng, 61, (1.0, 0.9, 0.5, 0.25, 0.1), 700)
    refs.append(np.zeros(0, np.uint64))
    values, offsets = synth.pack(refs)
    sample = synth.sample_from_refs(rng, refs, [0, 4, 150, 151, 152, 300], 0.6, 60000)
    return refs, values, offsets, sample


def test_two_shards_simulated_in_one_process(hip_lib):
    """References cut into two shards (a cluster straddles the cut); pairs routed to two hash-range
    posting handles by hand (what ShardedRun's all_to_all does); partial sums added; result equal
    to the oracle on the whole database."""
    import torch

    refs, values, offsets, sample = _data()
    n = len(refs)
    want_ov = oracle.overlap(values, offsets, sample)
    want_e, want_m = oracle.exclusive(values, offsets, want_ov > 0, sample)

    dev = torch.device("cuda:0")
    plan = ydist.shard_plan(offsets, 2)
    max_hash = int(values.max())
    bound = (max_hash + 1) // 2
    s_t = torch.from_numpy(sample.view(np.int64).copy()).to(dev)

    overlap = torch.zeros(n, dtype=torch.int32, device=dev)
    pairs = []
    shards = []
    pshift = 64
    for (b, e) in plan:
        v, o = ydist.slice_csr(values, offsets, b, e)
        v_t = torch.from_numpy(v.view(np.int64).copy()).to(dev)
        o_t = torch.from_numpy(o.view(np.int64).copy()).to(dev)
        db = RefDB.from_device(v_t.data_ptr(), o_t.data_ptr(), e - b, flags=YH_DB_NO_INDEX)
        shards.append((db, v_t, o_t))
        pshift = min(pshift, db.info()["partition_shift"])
        out = torch.zeros(e - b, dtype=torch.int32, device=dev)
        db.overlap_device(s_t.data_ptr(), s_t.numel(), out.data_ptr())
        db.synchronize()
        overlap[b:e] = out
        sizes = (o_t[1:] - o_t[:-1])
        ids = torch.repeat_interleave(torch.arange(e - b, device=dev, dtype=torch.int32) + b, sizes)
        pairs.append((v_t, ids))
    assert np.array_equal(overlap.cpu().numpy().view(np.uint32), want_ov)

    all_h = torch.cat([p[0] for p in pairs])
    all_r = torch.cat([p[1] for p in pairs])
    lo_sel = (all_h ^ SIGN) < (bound - 2 ** 63)
    sums = torch.zeros((3, n), dtype=torch.int32, device=dev)
    nshared = torch.zeros(n, dtype=torch.int32, device=dev)
    mask = (overlap != 0).to(torch.uint8).contiguous()
    posts = []
    for sel in (lo_sel, ~lo_sel):
        h = all_h[sel].contiguous()
        r = all_r[sel].contiguous()
        torch.cuda.synchronize()
        post = RefDB.from_pairs(h.data_ptr(), r.data_ptr(), h.numel(), n, pshift, max_hash)
        posts.append((post, h, r))
        part = torch.zeros((3, n), dtype=torch.int32, device=dev)
        post.exclusive_partial_device(mask.data_ptr(), s_t.data_ptr(), s_t.numel(), part[0].data_ptr(),
                                      part[1].data_ptr(), part[2].data_ptr())
        ns = torch.zeros(n, dtype=torch.int32, device=dev)
        post.nshared_device(ns.data_ptr())
        post.synchronize()
        sums += part
        nshared += ns
    sizes_all = torch.from_numpy(np.diff(offsets).astype(np.int32)).to(dev)
    e_t = torch.zeros(n, dtype=torch.int32, device=dev)
    m_t = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    posts[0][0].exclusive_finalize_device(n, mask.data_ptr(), sizes_all.data_ptr(), nshared.data_ptr(),
                                          overlap.data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(),
                                          sums[2].data_ptr(), e_t.data_ptr(), m_t.data_ptr())
    posts[0][0].synchronize()
    assert np.array_equal(e_t.cpu().numpy().view(np.uint32), want_e)
    assert np.array_equal(m_t.cpu().numpy().view(np.uint32), want_m)
    # posting-only handles refuse the streaming queries instead of answering wrongly
    from yacht_amd._lib import YachtHipError

    with pytest.raises(YachtHipError):
        posts[0][0].overlap(sample)
    for post, _h, _r in posts:
        post.close()
    for db, _v, _o in shards:
        db.close()


def test_sharded_run_hip_backend_world1(hip_lib):
    """ShardedRun end to end on the HIP backend (RCCL, one rank): equal to the single-handle path."""
    import torch
    import torch.distributed as dist

    refs, values, offsets, sample = _data()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        v_t = torch.from_numpy(values.view(np.int64).copy()).to(dev)
        o_t = torch.from_numpy(offsets.view(np.int64).copy()).to(dev)
        s_t = torch.from_numpy(sample.view(np.int64).copy()).to(dev)
        run = ydist.ShardedRun(v_t, o_t, ydist.HipBackend(0))
        ov, e, m = run.run(s_t)
        with RefDB(values, offsets) as db:
            wov, we, wm = db.run_counts(sample)
        assert np.array_equal(ov.cpu().numpy().view(np.uint32), wov)
        assert np.array_equal(e.cpu().numpy().view(np.uint32), we)
        assert np.array_equal(m.cpu().numpy().view(np.uint32), wm)
        run.close()
    finally:
        dist.destroy_process_group()
