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
from yacht_amd.engine import RefDB, YH_DB_FULL_INDEX, YH_DB_KEEP_CSR  # noqa: E402

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
with RefDB(values, offsets, flags=YH_DB_KEEP_CSR | YH_DB_FULL_INDEX) as db:
    import torch

    for it in range(rounds):
        k = int(rng.integers(len(samples)))
        s = samples[k]
        ov, e, m = want[k]
        op = int(rng.integers(6))
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
print("mixed calls ok:", rounds, "rounds")
