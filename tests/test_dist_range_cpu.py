"""HashRangeRefDB (yacht_amd/dist.py: the run step with the HASH SPACE spread over the ranks) on CPU: world_size 2 and 3,
gloo.  The plumbing -- range bounds, cutting every reference and every sample to a range, the bit rows of a block of
samples through one all-gather, the OR over the ranks, the sum of the shares -- is the code under test; the per-rank
arithmetic is stood in by a set-based restatement of the two library calls (test infrastructure).  The reduced result
must equal the oracle on the WHOLE database."""
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


class SetRangeBackend:
    """yh_run_local_range_device / yh_run_finish_range_device restated with Python sets."""

    def make_range_db(self, values_t, offsets_t):
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

        class L:
            handle = None

            def run_local(self, sample_t, a, b, counts_t, bits_t, ctx=0):
                S = set(sample_t.numpy().view(np.uint64)[a:b].tolist())
                ov = np.array([len(S & r) for r in refs], dtype=np.int64)
                sh = np.array([sum(1 for h in (S & r) if h in shared) for r in refs], dtype=np.int64)
                counts_t[0] = torch.from_numpy(ov.astype(np.int32))
                counts_t[2] = torch.from_numpy((ov - sh).astype(np.int32))
                bits = np.zeros(bits_t.numel(), dtype=np.uint32)
                for j in np.flatnonzero(ov > 0):
                    bits[j >> 5] |= np.uint32(1 << (j & 31))
                bits_t.copy_(torch.from_numpy(bits.view(np.int32)))

            def run_finish(self, gathered_t, n_ranks, stride_words, counts_t, ctx=0):
                g = gathered_t.numpy().view(np.uint32)
                words = (len(refs) + 31) // 32
                acc = np.zeros(words, dtype=np.uint32)
                for r in range(n_ranks):
                    acc |= g[r * stride_words: r * stride_words + words]
                mask = [bool((acc[j >> 5] >> (j & 31)) & 1) for j in range(len(refs))]
                e = np.zeros(len(refs), dtype=np.int32)
                for j, r in enumerate(refs):
                    if mask[j]:
                        e[j] = len(r) - nshared[j] + sum(1 for h in r if h in shared and not any(mask[x] for x in shared[h] if x != j))
                counts_t[1] = torch.from_numpy(e)

            def batch_local(self, cat_t, soff_t, n_samples, total, ov_t, words_t, slot=0):
                cat = cat_t.numpy().view(np.uint64)
                off = soff_t.numpy()
                N_ = len(refs)
                words = np.zeros(((n_samples + 63) // 64) * N_, dtype=np.uint64)  # planes of 64 samples: word [(s >> 6) * N + r], bit s & 63
                self._slots = getattr(self, "_slots", {})
                self._hit = self._slots.setdefault(slot, {})["hit"] = []
                for s in range(n_samples):
                    S = set(cat[off[s]:off[s + 1]].tolist())
                    self._hit.append(S)
                    ov = np.array([len(S & r) for r in refs], dtype=np.int32)
                    ov_t[s] = torch.from_numpy(ov)
                    words[(s >> 6) * N_: (s >> 6) * N_ + N_] |= (ov > 0).astype(np.uint64) << np.uint64(s & 63)
                words_t.view(-1)[: words.size].copy_(torch.from_numpy(words.view(np.int64)))

            def batch_finish(self, n_samples, gathered_t, n_ranks, ov_t, e_t, m_t, slot=0):
                N_ = len(refs)
                nw = ((n_samples + 63) // 64) * N_
                words = np.bitwise_or.reduce(gathered_t.numpy().view(np.uint64).reshape(gathered_t.shape[0], -1)[:n_ranks, :nw], axis=0)
                self._slots[slot]["words"] = words.copy()
                self._slots[slot]["n"] = n_samples
                hit = self._slots[slot]["hit"]
                for s in range(n_samples):
                    mask = ((words[(s >> 6) * N_: (s >> 6) * N_ + N_] >> np.uint64(s & 63)) & np.uint64(1)).astype(bool)
                    S = hit[s]
                    e = np.zeros(len(refs), dtype=np.int32)
                    m = np.zeros(len(refs), dtype=np.int32)
                    for j, r in enumerate(refs):
                        if mask[j]:
                            excl = [h for h in r if h not in shared or not any(mask[x] for x in shared[h] if x != j)]
                            e[j] = len(excl)
                            m[j] = sum(1 for h in excl if h in S)
                    e_t[s] = torch.from_numpy(e)
                    m_t[s] = torch.from_numpy(m)

            def words_pack(self, words_t, packed_t, cap):  # yh_run_batch_words_pack_device restated
                w = words_t.numpy().view(np.uint64)
                nz = np.flatnonzero(w)[::-1]  # (any order is allowed: take a different one than the kernel's)
                out = np.zeros(packed_t.numel(), dtype=np.uint64)
                out[0] = nz.size
                k = min(int(nz.size), int(cap))
                out[1:1 + k] = w[nz[:k]]
                out[1 + cap:].view(np.uint32)[:k] = nz[:k].astype(np.uint32)
                packed_t.copy_(torch.from_numpy(out.view(np.int64)))

            def words_unpack(self, gathered_t, n_ranks, cap, words_out_t, overflow_t):
                g = gathered_t.numpy().view(np.uint64).reshape(n_ranks, -1)
                acc = np.zeros(words_out_t.numel(), dtype=np.uint64)
                ov = 0
                for k in range(n_ranks):
                    n = int(g[k, 0])
                    ov |= int(n > cap)
                    n = min(n, int(cap))
                    ids = g[k, 1 + cap:].view(np.uint32)[:n]
                    np.bitwise_or.at(acc, ids, g[k, 1:1 + n])
                words_out_t.view(-1).copy_(torch.from_numpy(acc.view(np.int64)))
                overflow_t[0] = ov

            def _entries(self, slot):  # (reference, sample) of every set bit of the slot's global words, in that order
                words, n = self._slots[slot]["words"], self._slots[slot]["n"]
                return [(r, s_) for r in range(len(refs)) for s_ in range(n) if (int(words[(s_ >> 6) * len(refs) + r]) >> (s_ & 63)) & 1]

            def rows_pack(self, counts_t, vals_t, nrows_t, slot=0):
                ent = self._entries(slot)
                nrows_t[0] = len(ent)
                for k, (r, s_) in enumerate(ent[: vals_t.shape[0]]):
                    for c in range(3):
                        vals_t[k, c] = counts_t[c, s_, r]

            def rows_unpack(self, vals_t, rows_t, nrows_t, slot=0):
                ent = self._entries(slot)
                nrows_t[0] = len(ent)
                for k, (r, s_) in enumerate(ent[: vals_t.shape[0]]):
                    rows_t[k] = torch.tensor([s_, r, int(vals_t[k, 0]), int(vals_t[k, 1]), int(vals_t[k, 2])], dtype=torch.int32)

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
        big = np.unique(rng.integers(2 ** 63, 2 ** 64 - 1, size=200, dtype=np.uint64))  # hashes above 2^63 (scaled = 1)
        refs[3] = np.union1d(refs[3], big)
        refs[len(refs) - 2] = np.union1d(refs[len(refs) - 2], big[:120])
        values, offsets = synth.pack(refs)
        vt = torch.from_numpy(values.view(np.int64).copy())
        ot = torch.from_numpy(offsets.astype(np.int64))
        bounds = ydist.hash_range_bounds(int(values.max()), world)
        v, o = ydist.slice_to_hash_range(vt, ot, bounds[rank], bounds[rank + 1])
        hr = ydist.HashRangeRefDB(v, o, bounds, SetRangeBackend())

        def check(full, sample, what):
            full = full.numpy().view(np.uint32)
            want_ov = oracle.overlap(values, offsets, sample)
            want_e, want_m = oracle.exclusive(values, offsets, want_ov > 0, sample)
            assert np.array_equal(full[0], want_ov), f"rank {rank}: overlap ({what})"
            assert np.array_equal(full[2], want_m), f"rank {rank}: n_match ({what})"
            assert np.array_equal(full[1], want_e), f"rank {rank}: n_excl ({what})"

        cases = [synth.sample_from_refs(rng, refs, [0, 3, 57, 58, 100], 0.6, 4000),
                 synth.sample_from_refs(rng, refs, list(range(0, len(refs), 2)), 0.5, 3000),
                 np.unique(rng.integers(0, 2 ** 40, size=500, dtype=np.uint64)),     # all of it in the first range
                 np.zeros(0, np.uint64),
                 np.union1d(big[::3], refs[0][:5])]                                # the two ends of the hash space
        for k, sample in enumerate(cases):
            st = torch.from_numpy(sample.view(np.int64).copy())
            check(hr.gather(hr.run(st)), sample, f"case {k}")
        # one staging tensor refilled IN PLACE with another sample of the same length (nothing may be
        # remembered per tensor address -- the second sample's span in this rank's range differs from the first's)
        n_st = min(cases[0].size, cases[1].size)
        stage = torch.zeros(n_st, dtype=torch.int64)
        for k in (0, 1, 0):
            stage.copy_(torch.from_numpy(cases[k][:n_st].view(np.int64).copy()))
            check(hr.gather(hr.run(stage)), cases[k][:n_st], f"refilled staging tensor, case {k}")
        # a reference whose ONLY overlap lies in another rank's range still joins this rank's subset: its exclusive
        # count here must see it -- case 4 above (hashes at the two ends) is that situation for the middle rank.
        # blocks of 3 samples per exchange, two blocks in flight, a partly filled block, one reduce per block
        hr3 = ydist.HashRangeRefDB(v, o, bounds, SetRangeBackend(), block=3)
        samples = [synth.sample_from_refs(rng, refs, [int(x) for x in rng.choice(len(refs), size=4 + k, replace=False)], 0.5, 2000)
                   for k in range(5)]
        ts = [torch.from_numpy(x.view(np.int64).copy()) for x in samples]
        blk = [torch.zeros((3, 3, hr3.n_total), dtype=torch.int32) for _ in range(2)]
        for g in range(3):
            hr3.begin(ts[g], blk[0][g], 0, g)
        hr3.exchange(0)
        for g in range(2):
            hr3.begin(ts[3 + g], blk[1][g], 1, g)
        hr3.exchange(1)
        for g in range(3):
            hr3.end(blk[0][g], 0, g)
        for g in range(2):
            hr3.end(blk[1][g], 1, g)
        tot = [hr3.reduce(b.clone(), dst=0) for b in blk]  # the block's rows in ONE collective, to rank 0
        if rank == 0:
            for k, smp in enumerate(samples):
                check(tot[k // 3][k % 3], smp, f"blocked {k}")
        # the batched form: all five samples (+ an empty one) in ONE pass around one exchange of their subset words
        batch = samples + [np.zeros(0, np.uint64)]
        cb = hr.run_batch([torch.from_numpy(x.view(np.int64).copy()) for x in batch])
        full = hr.reduce(cb.clone())
        for k, smp in enumerate(batch):
            check(full[:, k, :], smp, f"batched {k}")
        # the result path in compact form (BatchRowsReducer): three blocks in flight in three batch slots, value triples
        # summed to rank 0; the second reducer starts with a collective too small for its blocks -> the dense fallback
        # for that block, a larger collective afterwards -- every block against the oracle on the whole database
        blocks = [samples[:3], samples[3:] + [np.zeros(0, np.uint64)], samples[1:4], samples[:2]]
        tb = [[torch.from_numpy(x.view(np.int64).copy()) for x in blk_] for blk_ in blocks]
        for first_cap in (None, 4):
            red = ydist.BatchRowsReducer(hr, batch=3, dst=0, nbuf=3, cap_rows=first_cap)
            cnt = [torch.zeros((3, 3, hr.n_total), dtype=torch.int32) for _ in range(3)]
            wrd = [torch.zeros(hr.n_total, dtype=torch.int64) for _ in range(3)]
            gth = [torch.zeros((world, hr.n_total), dtype=torch.int64) for _ in range(3)]
            got = {}

            def finish(j):
                rows, dense = red.finish(j % 3)
                if rank == 0:
                    nb = len(blocks[j])
                    got[j] = dense[:, :nb] if rows is None else ydist.BatchRowsReducer.rows_to_dense(rows, nb, hr.n_total)

            prev = None
            for j, blk_ in enumerate(tb):
                b = j % 3
                if j >= 3:
                    finish(j - 3)
                hr.batch_begin(hr.pack_batch(blk_), cnt[b], wrd[b], slot=b)
                hr.batch_exchange(wrd[b], gth[b])
                if prev is not None:  # the previous block's second half behind this block's exchange
                    pj, pb = prev
                    hr.batch_end(len(tb[pj]), gth[pb], cnt[pb], slot=pb)
                    red.send(pb, len(tb[pj]), cnt[pb], slot=pb)
                prev = (j, b)
            pj, pb = prev
            hr.batch_end(len(tb[pj]), gth[pb], cnt[pb], slot=pb)
            red.send(pb, len(tb[pj]), cnt[pb], slot=pb)
            for j in range(max(0, len(tb) - 3), len(tb)):
                finish(j)
            if first_cap is not None:
                assert red.n_overflow >= 1 and red.cap > first_cap, "the undersized collective must have been noticed"
            else:
                assert red.n_overflow == 0
            if rank == 0:
                for j, blk_ in enumerate(blocks):
                    for k, smp in enumerate(blk_):
                        check(got[j][:, k, :], smp, f"compact rows, first_cap {first_cap}, block {j} sample {k}")
        # the whole pipeline as the product drives it (BatchedRangeRunner): compact subset words + compact rows, three blocks
        # in flight; undersized exchanges of either kind are noticed on every rank, the block repeated, the capacity raised
        for kw, expect in ((dict(), "fits"), (dict(cap_words=2), "words"), (dict(cap_rows=4), "rows"),
                           (dict(cap_words=3, cap_rows=5), "both"), (dict(compact_words=False), "fits"),
                           (dict(dense_rows=True, cap_words=2), "words"),
                           # fewer slots: one slot = the two halves of successive blocks in turn, no overlap
                           (dict(nbuf=1), "fits"), (dict(nbuf=2), "fits"), (dict(nbuf=1, cap_words=2, cap_rows=4), "both"),
                           (dict(nbuf=1, dense_rows=True), "fits")):
            got = {}

            def on_result(tag, n_in, rows, dense):
                if rank == 0:
                    got[tag] = dense[:, :n_in].clone() if rows is None else ydist.BatchRowsReducer.rows_to_dense(rows, n_in, hr.n_total)
                else:
                    assert rows is None and dense is None

            run = ydist.BatchedRangeRunner(hr, batch=3, dst=0, on_result=on_result, **dict(dict(nbuf=3), **kw))
            for rep in range(2):  # (the second round starts with the raised capacities)
                for j, blk_ in enumerate(tb):
                    run.submit(hr.pack_batch(blk_), len(blk_), tag=(rep, j))
                run.drain()
            if expect in ("words", "both"):
                assert run.n_words_overflow >= 1 and run.cap_words > kw["cap_words"], "the undersized word exchange must have been noticed"
            else:
                assert run.n_words_overflow == 0
            if expect in ("rows", "both"):
                assert run.red.n_overflow >= 1
            cb_ = run.collective_bytes()
            assert cb_["total"] == cb_["subset_words_all_gather"] + cb_["result"]
            if rank == 0:
                assert sorted(got) == [(rep, j) for rep in range(2) for j in range(len(tb))]
                for (rep, j), dense in got.items():
                    for k, smp in enumerate(blocks[j]):
                        check(dense[:, k, :], smp, f"runner {kw} round {rep} block {j} sample {k}")
        # blocks of MORE than 64 samples (round 6: subset words in planes of 64 samples, up to 256 per block): 70, 130 and 5
        # samples through a runner of batch 130 -- entries of planes 1 and 2, a partly filled last plane, an undersized exchange
        big = [[samples[k % 5] for k in range(70)], [samples[(3 * k + 1) % 5] if k % 7 else np.zeros(0, np.uint64) for k in range(130)], samples]
        tbig = [[torch.from_numpy(x.view(np.int64).copy()) for x in blk_] for blk_ in big]
        for kw in (dict(), dict(cap_words=5, cap_rows=9), dict(compact_words=False, nbuf=2)):
            got = {}

            def on_big(tag, n_in, rows, dense):
                if rank == 0:
                    got[tag] = dense[:, :n_in].clone() if rows is None else ydist.BatchRowsReducer.rows_to_dense(rows, n_in, hr.n_total)

            run = ydist.BatchedRangeRunner(hr, batch=130, dst=0, on_result=on_big, **kw)
            assert run.planes == 3 and run.words[0].numel() == 3 * hr.n_total
            for j, blk_ in enumerate(tbig):
                run.submit(hr.pack_batch(blk_), len(blk_), tag=j)
            run.drain()
            if "cap_words" in kw:
                assert run.n_words_overflow >= 1 and run.red.n_overflow >= 1
            if rank == 0:
                for j, blk_ in enumerate(big):
                    for k in (0, 1, 63, 64, 65, 69, 127, 128, 129):
                        if k < len(blk_):
                            check(got[j][:, k, :], blk_[k], f"big blocks {kw} block {j} sample {k}")
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_hash_range_refdb_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_bounds_and_slices():
    import torch

    rng = np.random.default_rng(0)
    refs = [np.unique(rng.integers(0, 2 ** 64 - 1, size=200, dtype=np.uint64)) for _ in range(7)]
    refs[3] = np.zeros(0, np.uint64)
    values, offsets = synth.pack(refs)
    vt = torch.from_numpy(values.view(np.int64).copy())
    ot = torch.from_numpy(offsets.astype(np.int64))
    for world in (1, 2, 3, 8):
        b = ydist.hash_range_bounds(int(values.max()), world)
        assert b[0] == 0 and b[-1] == 2 ** 64 and all(b[i] < b[i + 1] for i in range(world))
        total = 0
        sample = np.unique(rng.integers(0, 2 ** 64 - 1, size=1000, dtype=np.uint64))
        st = torch.from_numpy(sample.view(np.int64).copy())
        covered = 0
        for g in range(world):
            v, o = ydist.slice_to_hash_range(vt, ot, b[g], b[g + 1])
            v, o = v.numpy().view(np.uint64), o.numpy()
            for j, r in enumerate(refs):
                inr = (r >= np.uint64(b[g])) & ((r < np.uint64(b[g + 1])) if b[g + 1] < 2 ** 64 else np.ones(r.size, bool))
                assert np.array_equal(v[o[j]:o[j + 1]], r[inr])
            total += v.size
            a, e = ydist.sample_slice(st, b[g], b[g + 1])
            assert a == covered
            covered = e
        assert total == values.size and covered == sample.size
