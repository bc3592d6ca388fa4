#!/usr/bin/env python3
"""Rank 0's share of a G-way hash-range run, blocks of B samples (default 64; round 6: up to 256) through dist.BatchedRangeRunner (what bench.py's scaling_model
times) -- alone in a process, for rocprofv3 --kernel-trace --stats: which kernels the ~0.17 ms per block that do not shrink with G
are made of.   usage: rocprofv3 --kernel-trace --stats -d /tmp/p -- python3 scripts/probes/batch_share_trace.py 8 [blocks] [B]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from yacht_amd import dist as ydist, synth  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda", 0)
n = 85_205
plan = synth.global_db_plan(1002, n, cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
values, offsets = synth.global_db_refs_device(plan, np.arange(n), device="cuda:0")
samples = [synth.global_db_sample_device(plan, 2002 + i, n_sample=1_000_000, n_present=200, device="cuda:0") for i in range(B)]
mh = synth.max_hash_for_scaled(1000)
bg = ydist.hash_range_bounds(mh, G)
v_g, o_g = (values, offsets) if G == 1 else ydist.slice_to_hash_range(values, offsets, bg[0], bg[1])
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    hr = ydist.HashRangeRefDB(v_g, o_g, [bg[0], bg[1]], ydist.HipRangeBackend(0), block=1)
    hr.local.handle.set_stream(stream.cuda_stream)
    packed = hr.pack_batch(samples)
    run = ydist.BatchedRangeRunner(hr, batch=B, dst=0, nbuf=3)
    for _ in range(3):
        run.submit(packed, B)
    run.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(blocks):
        run.submit(packed, B)
    run.drain()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("G = %d: %.4f ms per block of %d (%.2f us per sample)   word-exchange overflows %d, rows overflows %d" % (G, 1e3 * el / blocks, B, 1e6 * el / blocks / B, run.n_words_overflow, run.red.n_overflow), file=sys.stderr)
hr.close()
