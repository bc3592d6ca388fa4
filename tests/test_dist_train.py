"""`yacht train` over ranks (BASELINE.json configs[3], "tiled across GPUs"): row blocks of the pairwise
matrix per rank, pair lists all-gathered, selection on the concatenated list.  CPU: world_size 2 over
gloo, the per-block compute stood in by the oracle.  GPU: the same plumbing over RefDB.pairwise."""
import os
import socket

import numpy as np
import pytest

from yacht_amd import dist as ydist
from yacht_amd import synth


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    values, offsets = synth.config4(seed=77, n_clusters=30, size=300)
    return values, offsets, 0.95 ** 31


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import torch.distributed as dist

    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        values, offsets, c = _case()
        sizes = np.diff(offsets).astype(np.uint32)
        wi, wj, wc, _ = oracle.train_pairs(values, offsets, c)
        # work per row ~ shared hashes of the row's reference; any non-negative weights give a valid plan
        nshared = np.bincount(wi, minlength=sizes.size)
        plan = ydist.pair_row_plan(nshared, world)
        assert plan[0][0] == 0 and plan[-1][1] == sizes.size

        def rows(b, e):  # the rank's block of the oracle's list (the product calls RefDB.pairwise(c, b, e))
            keep = (wi >= b) & (wi < e)
            return wi[keep], wj[keep], wc[keep]

        gi, gj, gc = ydist.sharded_pairwise(rows, plan)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
        assert np.array_equal(oracle.train_select(sizes, gi, gj), oracle.train_select(sizes, wi, wj))
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_pair_row_plan_shapes():
    for world in (1, 2, 3, 8):
        plan = ydist.pair_row_plan(np.array([0, 10, 0, 0, 500, 3, 3, 0], dtype=np.uint32), world)
        assert plan[0][0] == 0 and plan[-1][1] == 8
        assert all(plan[r][1] == plan[r + 1][0] for r in range(world - 1))
    assert ydist.pair_row_plan(np.zeros(0, np.uint32), 2) == [(0, 0), (0, 0)]


def test_two_rank_train_gloo(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


@pytest.mark.gpu
def test_row_blocks_on_the_hip_engine(hip_lib):
    """Row blocks computed by RefDB.pairwise, stitched like the ranks would, equal the oracle."""
    import torch.distributed as dist

    from oracle import oracle
    from yacht_amd.engine import RefDB, train_select

    values, offsets, c = _case()
    sizes = np.diff(offsets).astype(np.uint32)
    wi, wj, wc, _ = oracle.train_pairs(values, offsets, c)
    with RefDB(values, offsets) as db:
        nshared = np.bincount(db.pairwise(c)[0], minlength=sizes.size)
        for world in (2, 5):
            plan = ydist.pair_row_plan(nshared, world)
            parts = [db.pairwise(c, b, e) for b, e in plan]
            gi, gj, gc = (np.concatenate([p[k] for p in parts]) for k in range(3))
            assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
        # and through the collective with one rank
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("gloo", rank=0, world_size=1)
        try:
            gi, gj, gc = ydist.sharded_pairwise(lambda b, e: db.pairwise(c, b, e), ydist.pair_row_plan(nshared, 1))
        finally:
            dist.destroy_process_group()
    assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
    assert np.array_equal(train_select(sizes, gi, gj), oracle.train_select(sizes, wi, wj))


@pytest.mark.gpu
def test_pairwise_only_handle(hip_lib):
    """YH_DB_PAIRWISE_ONLY: same pairs and index statistics as a full handle, no sample queries."""
    from oracle import oracle
    from yacht_amd import _lib
    from yacht_amd.engine import YH_DB_PAIRWISE_ONLY, RefDB

    values, offsets, c = _case()
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c)
    with RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY) as db:
        gi, gj, gc = db.pairwise(c)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc)
        assert db.index_stats() == wstats
        with pytest.raises(_lib.YachtHipError):
            db.overlap(np.unique(values)[:100])
        with pytest.raises(_lib.YachtHipError):
            db.run_counts(np.unique(values)[:100])
