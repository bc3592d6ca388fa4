#!/usr/bin/env python3
"""One handle, every query entry point in random order: the accumulators that are 'zero at rest'
(replica counters, shared-hash flags, exclusive sums) must stay consistent whatever ran before.

    python tests/tools/mix_calls.py [rounds] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle  # noqa: E402  (the checker)
from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import BATCH_ROW_DTYPE, ROW_DTYPE, RefDB, YH_DB_DEFAULT, YH_DB_KEEP_CSR  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
values, offsets, _ = synth.config3_like(seed=21, n_refs=3000, n_sample=10_000, n_present=10)
refs = [values[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]
n = len(refs)
samples = []
for k in range(6):
    present = rng.choice(n, size=int(rng.choice([0, 3, 40, 400])), replace=False)
    parts = [refs[i][rng.random(refs[i].size) < 0.6] for i in present]
    noise = np.unique(rng.integers(0, synth.max_hash_for_scaled(1000), size=int(rng.choice([0, 500, 20000])), dtype=np.uint64))
    samples.append(np.unique(np.concatenate(parts + [noise])) if parts else noise)
want = []
for s in samples:
    ov = oracle.overlap(values, offsets, s)
    e, m = oracle.exclusive(values, offsets, ov > 0, s)
    want.append((ov, e, m))
with RefDB(values, offsets, flags=YH_DB_KEEP_CSR | YH_DB_DEFAULT) as db:
    import torch

    for it in range(rounds):
        k = int(rng.integers(len(samples)))
        s = samples[k]
        ov, e, m = want[k]
        op = int(rng.integers(9))
        if op == 0:
            got = db.run_counts(s)
            assert all(np.array_equal(a, b) for a, b in zip(got, (ov, e, m))), ("run_counts", it)
        elif op == 1:
            assert np.array_equal(db.overlap(s), ov), ("overlap", it)
        elif op == 2:
            mask = rng.random(n) < 0.3
            we, wm = oracle.exclusive(values, offsets, mask, s)
            ge, gm = db.exclusive(mask, s)
            assert np.array_equal(ge, we) and np.array_equal(gm, wm), ("exclusive", it)
        elif op == 3:
            assert np.array_equal(db.overlap(s, method="bsearch"), ov), ("bsearch", it)
        elif op == 4 and s.size:
            d_s = torch.from_numpy(s.view(np.int64)).cuda()
            out = torch.zeros((3, n), dtype=torch.int32, device="cuda")
            db.run_indexed_device(d_s.data_ptr(), s.size, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
            db.synchronize()
            g = out.cpu().numpy().view(np.uint32)
            assert np.array_equal(g[0], ov) and np.array_equal(g[1], e) and np.array_equal(g[2], m), ("indexed run", it)
        elif op == 5 and s.size:
            d_s = torch.from_numpy(s.view(np.int64)).cuda()
            out = torch.zeros(n, dtype=torch.int32, device="cuda")
            db.overlap_device(d_s.data_ptr(), s.size, out.data_ptr())
            db.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint32), ov), ("overlap_device", it)
        elif op == 6 and s.size:
            # a PIPELINED step, then the compact rows of it with no join in between -- the rows entry point
            # has to run the pending reducer / exclusive stages first and read the bits of the step's own context
            d_s = torch.from_numpy(s.view(np.int64)).cuda()
            out = torch.zeros((3, n), dtype=torch.int32, device="cuda")
            rows = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
            nrows = torch.zeros(1, dtype=torch.int32, device="cuda")
            for _ in range(int(rng.integers(1, 4))):  # (one to three steps: every rotation of the three contexts)
                db.run_device_pipelined(d_s.data_ptr(), s.size, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
            db.run_rows_device(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), rows.data_ptr(), n, nrows.data_ptr())
            db.synchronize()
            k_rows = int(nrows.item())
            r = rows[:k_rows].cpu().numpy().view(np.uint32)
            hit = np.flatnonzero(ov > 0)
            assert k_rows == hit.size and np.array_equal(r[:, 0], hit) and np.array_equal(r[:, 1], ov[hit]) and \
                np.array_equal(r[:, 2], e[hit]) and np.array_equal(r[:, 3], m[hit]), ("rows after a pipelined step", it)
        elif op == 7 and s.size:
            # a pipelined step left pending, then the two halves of a sharded step in one of the contexts 0..2 the
            # pending stages still read: both results must be right
            d_s = torch.from_numpy(s.view(np.int64)).cuda()
            k2 = int(rng.integers(len(samples)))
            s2 = samples[k2]
            if s2.size:
                d_s2 = torch.from_numpy(s2.view(np.int64)).cuda()
                out = torch.zeros((3, n), dtype=torch.int32, device="cuda")
                out2 = torch.zeros((3, n), dtype=torch.int32, device="cuda")
                bits = torch.zeros(((n + 255) // 256) * 8, dtype=torch.int32, device="cuda")
                db.run_device_pipelined(d_s.data_ptr(), s.size, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
                ctx = int(rng.integers(3))
                db.run_local_device(d_s2.data_ptr(), s2.size, out2[0].data_ptr(), out2[1].data_ptr(), out2[2].data_ptr(), bits.data_ptr(), ctx)
                db.run_finish_device(bits.data_ptr(), out2[1].data_ptr(), ctx)
                db.synchronize()
                g = out.cpu().numpy().view(np.uint32)
                assert np.array_equal(g[0], ov) and np.array_equal(g[1], e) and np.array_equal(g[2], m), ("pipelined step before a sharded one", it)
                g2 = out2.cpu().numpy().view(np.uint32)
                w2 = want[k2]
                assert all(np.array_equal(g2[c], w2[c]) for c in range(3)), ("sharded halves after a pipelined step", it)
        elif op == 8:
            # a batch of three samples and its result in compact form (pack -> unpack on the one device)
            ks = [int(x) for x in rng.integers(len(samples), size=3)]
            cat = np.concatenate([samples[q] for q in ks]) if sum(samples[q].size for q in ks) else np.zeros(0, np.uint64)
            soff = np.concatenate([[0], np.cumsum([samples[q].size for q in ks])]).astype(np.int64)
            d_cat = torch.from_numpy(cat.view(np.int64).copy()).cuda() if cat.size else torch.zeros(1, dtype=torch.int64, device="cuda")
            d_off = torch.from_numpy(soff).cuda()
            out = torch.zeros((3, 3, n), dtype=torch.int32, device="cuda")
            db.run_batch_device(d_cat.data_ptr(), d_off.data_ptr(), 3, int(cat.size), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
            vals = torch.zeros((3 * n, 3), dtype=torch.int32, device="cuda")
            brow = torch.zeros((3 * n, 5), dtype=torch.int32, device="cuda")
            n1 = torch.zeros(1, dtype=torch.int32, device="cuda")
            n2 = torch.zeros(1, dtype=torch.int32, device="cuda")
            db.run_batch_rows_pack_device(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), vals.data_ptr(), 3 * n, n1.data_ptr())
            db.run_batch_rows_unpack_device(vals.data_ptr(), 3 * n, brow.data_ptr(), n2.data_ptr())
            db.synchronize()
            g = out.cpu().numpy().view(np.uint32)
            for t, q in enumerate(ks):
                assert all(np.array_equal(g[c][t], want[q][c]) for c in range(3)), ("batch", it, t)
            kr = int(n1.item())
            assert kr == int(n2.item()) == sum(int((want[q][0] > 0).sum()) for q in ks), ("batch rows: count", it)
            br = brow[:kr].cpu().numpy().view(np.uint32)
            order = np.lexsort((br[:, 0], br[:, 1]))
            assert np.array_equal(order, np.arange(kr)), ("batch rows: (ref, sample) order", it)
            for t, q in enumerate(ks):
                mine = br[br[:, 0] == t]
                hit = np.flatnonzero(want[q][0] > 0)
                assert np.array_equal(mine[:, 1], hit) and all(np.array_equal(mine[:, 2 + c], want[q][c][hit]) for c in range(3)), ("batch rows", it, t)
print("mixed calls ok:", rounds, "rounds")
