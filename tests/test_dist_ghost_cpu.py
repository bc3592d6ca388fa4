"""ShardedRefDB (ghost design, yacht_amd/dist.py) on CPU: world_size 2 and 3, gloo.  The exchange --
pairs to the hash-range owners, foreign postings back, ghost references, bit positions of the subset
exchange -- is the code under test; the per-rank arithmetic is stood in by a set-based restatement of
the two library calls (test infrastructure), and the gathered result must equal the oracle on the WHOLE
database."""
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


class SetLocalBackend:
    """yh_run_local_device / yh_run_finish_device restated with Python sets."""

    def make_local_db(self, values_t, offsets_t, ghost_begin, ghost_src_t):
        import torch

        v = values_t.numpy().view(np.uint64)
        o = offsets_t.numpy()
        refs = [set(v[o[j]:o[j + 1]].tolist()) for j in range(o.size - 1)]
        holders = {}
        for j, r in enumerate(refs):
            for h in r:
                holders.setdefault(h, []).append(j)
        shared = {h: js for h, js in holders.items() if len(js) > 1}
        nshared = [sum(1 for h in r if h in shared) for r in refs]
        gsrc = ghost_src_t.numpy().astype(np.int64)
        n = len(refs)
        state = {}

        class L:
            def run_local(self, sample_t, counts_t, bits_t, ctx=0):
                S = set(sample_t.numpy().view(np.uint64).tolist())
                ov = np.array([len(S & r) for r in refs], dtype=np.int64)
                sh = np.array([sum(1 for h in (S & r) if h in shared) for r in refs], dtype=np.int64)
                counts_t[0] = torch.from_numpy(ov.astype(np.int32))
                counts_t[2] = torch.from_numpy((ov - sh).astype(np.int32))
                counts_t[1] = torch.from_numpy(np.where(ov > 0, [len(r) - s for r, s in zip(refs, nshared)], 0).astype(np.int32))
                bits = np.zeros(bits_t.numel(), dtype=np.uint32)
                for j in np.flatnonzero(ov > 0):
                    bits[j >> 5] |= np.uint32(1 << (j & 31))
                bits_t.copy_(torch.from_numpy(bits.view(np.int32)))
                state[ctx] = ov > 0

            def run_finish(self, global_bits_t, counts_t, ctx=0):
                gb = global_bits_t.numpy().view(np.uint32)
                mask = state[ctx].copy()
                for k, b in enumerate(gsrc.tolist()):
                    mask[ghost_begin + k] = bool((gb[b >> 5] >> (b & 31)) & 1)
                e = counts_t[1].numpy().copy()
                for j in range(ghost_begin):
                    if mask[j]:
                        e[j] += sum(1 for h in refs[j] if h in shared and not any(mask[x] for x in shared[h] if x != j))
                counts_t[1] = torch.from_numpy(e)

            def close(self):
                pass

        return L()


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import torch
    import torch.distributed as dist

    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # (no host-name look-ups: they stall on some boxes)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        refs = synth.clustered_refs(rng, 23, (1.0, 0.9, 0.5, 0.25, 0.1), 300)
        refs[9] = np.zeros(0, np.uint64)
        # hashes above 2^63 (scaled = 1) in one cluster, and a hash held by references of every shard
        big = np.unique(rng.integers(2 ** 63, 2 ** 64 - 1, size=200, dtype=np.uint64))
        refs[3] = np.union1d(refs[3], big)
        refs[len(refs) - 2] = np.union1d(refs[len(refs) - 2], big[:120])
        everywhere = np.uint64(123456789)
        for j in (0, len(refs) // 2, len(refs) - 1):
            refs[j] = np.union1d(refs[j], [everywhere])
        values, offsets = synth.pack(refs)
        plan = ydist.shard_plan(offsets, world)
        b, e = plan[rank]
        v, o = ydist.slice_csr(values, offsets, b, e)
        sdb = ydist.ShardedRefDB(torch.from_numpy(v.view(np.int64).copy()), torch.from_numpy(o.astype(np.int64)),
                                 SetLocalBackend())
        for present, extra in (([0, 3, 57, 58, 100], []), (list(range(0, len(refs), 2)), [everywhere]), ([], [])):
            sample = synth.sample_from_refs(rng, refs, present, 0.6, 4000) if present else \
                np.unique(rng.integers(0, 2 ** 40, size=500, dtype=np.uint64))
            sample = np.union1d(sample, np.array(extra, dtype=np.uint64))
            st = torch.from_numpy(sample.view(np.int64).copy())
            if present and not extra:  # two steps in flight in the two contexts: begin(0), begin(1), end(0), end(1)
                other = np.unique(rng.integers(0, 2 ** 40, size=300, dtype=np.uint64))
                c0, c1 = sdb.new_counts(), sdb.new_counts()
                sdb.run_begin(st, c0, 0)
                sdb.run_begin(torch.from_numpy(other.view(np.int64).copy()), c1, 1)
                sdb.run_end(c0, 0)
                sdb.run_end(c1, 1)
                full = sdb.gather(c0).numpy().view(np.uint32)
                assert not sdb.gather(c1).numpy()[0].any()  # (the noise sample overlaps nothing)
            else:
                full = sdb.gather(sdb.run(st)).numpy().view(np.uint32)
            want_ov = oracle.overlap(values, offsets, sample)
            want_e, want_m = oracle.exclusive(values, offsets, want_ov > 0, sample)
            assert np.array_equal(full[0], want_ov), f"rank {rank}: overlap"
            assert np.array_equal(full[2], want_m), f"rank {rank}: n_match"
            assert np.array_equal(full[1], want_e), f"rank {rank}: n_excl"
        if world > 1:
            assert sdb.n_ghost > 0, "the cut was meant to go through a cluster"
        # blocks of 3 samples per bit exchange, two blocks in flight (what bench.py --gpus N drives): the local halves of
        # block 1 are queued before the second halves of block 0; a partly filled block at the end
        sdb3 = ydist.ShardedRefDB(torch.from_numpy(v.view(np.int64).copy()), torch.from_numpy(o.astype(np.int64)),
                                  SetLocalBackend(), block=3)
        samples = []
        for k in range(5):
            present = [int(x) for x in rng.choice(len(refs), size=4 + k, replace=False)]
            smp = synth.sample_from_refs(rng, refs, present, 0.5, 2000)
            samples.append(np.union1d(smp, [everywhere]) if k % 2 else smp)
        cs = [sdb3.new_counts() for _ in samples]
        ts = [torch.from_numpy(x.view(np.int64).copy()) for x in samples]
        for g in range(3):
            sdb3.begin(ts[g], cs[g], 0, g)
        sdb3.exchange(0)
        for g in range(2):
            sdb3.begin(ts[3 + g], cs[3 + g], 1, g)
        sdb3.exchange(1)
        for g in range(3):
            sdb3.end(cs[g], 0, g)
        for g in range(2):
            sdb3.end(cs[3 + g], 1, g)
        for smp, c in zip(samples, cs):
            full = sdb3.gather(c).numpy().view(np.uint32)
            want_ov = oracle.overlap(values, offsets, smp)
            want_e, want_m = oracle.exclusive(values, offsets, want_ov > 0, smp)
            assert np.array_equal(full[0], want_ov) and np.array_equal(full[2], want_m) and np.array_equal(full[1], want_e), \
                f"rank {rank}: blocked exchange"
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_refdb_ghosts_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
