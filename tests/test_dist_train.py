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
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (no host-name look-ups: they stall on some boxes)
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


def _range_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import torch.distributed as dist

    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (no host-name look-ups: they stall on some boxes)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        values, offsets, c = _case()
        sizes = np.diff(offsets).astype(np.uint32)
        wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c)
        bounds = ydist.hash_range_bounds(int(values.max()), world)
        v, o = ydist.slice_csr_to_hash_range(values, offsets, bounds[rank], bounds[rank + 1])

        def compute():  # (the product: RefDB(v, o, PAIRWISE_ONLY).pairwise(0.0) + index_stats())
            pi, pj, pc, st = oracle.train_pairs(v, o, 0.0)
            return pi, pj, pc, st

        gi, gj, gc, gstats = ydist.hash_range_pairwise(compute, sizes.size, sizes, c)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc), f"rank {rank}"
        assert tuple(gstats) == tuple(wstats)
        assert np.array_equal(oracle.train_select(sizes, gi, gj), oracle.train_select(sizes, wi, wj))
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_hash_range_train_gloo(tmp_path, world):
    """`yacht train` with the hash space over the ranks: per-range partial pair counts, one all-gather, sum, threshold on
    the totals -- equal to the oracle on the whole sketches (pairs, counts, the three statistics, the selection)."""
    import torch.multiprocessing as mp

    mp.spawn(_range_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_merge_partial_pairs_threshold_on_totals():
    sizes = np.array([1000, 2000, 10], dtype=np.uint32)
    a = (np.array([0, 1]), np.array([1, 0]), np.array([100, 100]))
    b = (np.array([0, 1, 2]), np.array([1, 0, 0]), np.array([150, 150, 3]))
    gi, gj, gc = ydist.merge_partial_pairs(3, [a, b], sizes, 0.25)
    # 250 / 1000 == 0.25 is kept (not below), 250 / 2000 is not, 3 / 10 is
    assert gi.tolist() == [0, 2] and gj.tolist() == [1, 0] and gc.tolist() == [250, 3]


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
        # (the collective itself runs in test_two_rank_train_gloo and, on the GPU, in test_bench_train_two_ranks_share_gpu:
        # a process group set up inside this process cost 14 minutes of host-name look-ups on one GPU box)
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


@pytest.mark.gpu
def test_hash_ranges_on_the_hip_engine(hip_lib):
    """`yacht train` by hash range on the HIP engine: per-range handles (YH_DB_PAIRWISE_ONLY over the range slices),
    yh_pairwise with c = 0 for the partial counts, summed and thresholded like the ranks would -- equal to the oracle on
    the whole sketches, statistics included; an empty range (more ranges than the hash space fills) is harmless."""
    from oracle import oracle
    from yacht_amd.engine import YH_DB_PAIRWISE_ONLY, RefDB, train_select

    values, offsets, c = _case()
    sizes = np.diff(offsets).astype(np.uint32)
    wi, wj, wc, wstats = oracle.train_pairs(values, offsets, c)
    for world in (1, 2, 5):
        bounds = ydist.hash_range_bounds(int(values.max()), world)
        parts, stats = [], np.zeros(3, dtype=np.int64)
        for g in range(world):
            v, o = ydist.slice_csr_to_hash_range(values, offsets, bounds[g], bounds[g + 1])
            with RefDB(v, o, flags=YH_DB_PAIRWISE_ONLY) as db:
                parts.append(db.pairwise(0.0))
                stats += np.array(db.index_stats(), dtype=np.int64)
        gi, gj, gc = ydist.merge_partial_pairs(sizes.size, parts, sizes, c)
        assert np.array_equal(gi, wi) and np.array_equal(gj, wj) and np.array_equal(gc, wc), world
        assert tuple(int(x) for x in stats) == tuple(wstats)
        assert np.array_equal(train_select(sizes, gi, gj), oracle.train_select(sizes, wi, wj))
    # a range above every hash: an empty handle
    v, o = ydist.slice_csr_to_hash_range(values, offsets, int(values.max()) + 1, 2 ** 64)
    with RefDB(v, o, flags=YH_DB_PAIRWISE_ONLY) as db:
        assert db.pairwise(0.0)[0].size == 0 and db.index_stats() == (0, 0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["hash", "rows"])
def test_bench_train_two_ranks_share_gpu(hip_lib, shard):
    """bench_train.py --gpus 2 over gloo on one GPU (both ways of splitting the work), bit-exact against the oracle and
    the genuine reference executable on its sample."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench_train.py"), "--gpus", "2", "--share-gpu",
                        "--backend", "gloo", "--shard", shard, "--clusters", "200", "--size", "600", "--steps", "2",
                        "--oracle-clusters", "200"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["parity_bit_exact"] is True and line["n_gpus"] == 2
    assert ("hash range" in line["scaling"]) == (shard == "hash")
