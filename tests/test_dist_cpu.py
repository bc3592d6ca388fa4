"""The N > 1 path on CPU: world_size 2, gloo.  The collective plumbing (shard plan, padding,
gather order) is the code under test; the per-shard compute is stood in by the CPU oracle."""
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


def _worker(rank: int, world: int, port: int, seed: int, out_dir: str) -> None:
    import torch.distributed as dist

    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (no host-name look-ups: they stall on some boxes)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)
        refs = synth.independent_refs(rng, 101, 400, 0.8, 0, 3000)  # ragged, some empty
        refs[17] = np.zeros(0, np.uint64)
        values, offsets = synth.pack(refs)
        sample = synth.sample_from_refs(rng, refs, [3, 50, 99, 100], 0.5, 5000)
        plan = ydist.shard_plan(offsets, world)
        b, e = plan[rank]
        v, o = ydist.slice_csr(values, offsets, b, e)
        got = ydist.sharded_overlap(sample, lambda s: oracle.overlap(v, o, s), plan)
        want = oracle.overlap(values, offsets, sample)
        assert np.array_equal(got, want), f"rank {rank}: gathered counts differ"
        # three count rows in one collective, as bench.py sends them
        import torch

        local = np.stack([oracle.overlap(v, o, sample), np.arange(e - b, dtype=np.uint32),
                          np.full(e - b, rank, np.uint32)]).view(np.int32)
        full = ydist.gather_counts(torch.from_numpy(local), plan).numpy().view(np.uint32)
        assert np.array_equal(full[0], want)
        assert np.array_equal(full[2], np.concatenate([np.full(pe - pb, r, np.uint32) for r, (pb, pe) in enumerate(plan)]))
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_shard_plan_balances_hashes():
    rng = np.random.default_rng(0)
    sizes = rng.integers(0, 5000, size=1000)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        plan = ydist.shard_plan(offsets, world)
        assert plan[0][0] == 0 and plan[-1][1] == 1000
        assert all(plan[r][1] == plan[r + 1][0] for r in range(world - 1))
        loads = [int(offsets[e] - offsets[b]) for b, e in plan]
        assert max(loads) - min(loads) <= 2 * 5000
    # degenerate: fewer references than ranks, empty database
    assert ydist.shard_plan(np.array([0, 5], np.uint64), 4)[-1][1] == 1
    assert ydist.shard_plan(np.array([0], np.uint64), 2) == [(0, 0), (0, 0)]


def test_two_rank_gather_gloo(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, 123, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
