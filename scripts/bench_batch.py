#!/usr/bin/env python3
"""Batched multi-sample `run` at bench.py's scale: B samples per call against
the GTDB-rs214-scale database (85 205 references, ~334 M hashes) held with its directory.

    python scripts/bench_batch.py [--batch 64] [--sample-hashes 200000] [--steps 20]

Prints one JSON line: batches timed with HIP events on the handle's stream, samples/s and
(sample, reference) queries/s, and a parity check of several rows of the batch against the
one-sample tile path on the same handle (itself pinned to the oracle by tests/ and bench.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_samples(torch, synth, vals, offsets, n_refs, batch, n_hashes, seed, dev):
    """`batch` samples: 200 references each at coverage Beta(0.5, 2), topped up with noise."""
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = []
    off_np = offsets.cpu().numpy()
    for _ in range(batch):
        parts = []
        for j in rng.choice(n_refs, size=200, replace=False):
            r = vals[int(off_np[j]):int(off_np[j + 1])]
            keep = torch.rand(r.numel(), generator=g, device=dev) < float(rng.beta(0.5, 2.0))
            parts.append(r[keep])
        have = sum(int(p.numel()) for p in parts)
        parts.append(torch.randint(0, synth.max_hash_for_scaled(1000), (max(n_hashes - have, 0),), generator=g, device=dev, dtype=torch.int64))
        out.append(torch.unique(torch.cat(parts)))
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--sample-hashes", type=int, default=200_000)
    ap.add_argument("--refs", type=int, default=85_205)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()

    import torch

    from yacht_amd import _lib, synth
    from yacht_amd.engine import RefDB

    if _lib.device_count() < 1 or not torch.cuda.is_available():
        print("needs an MI355X (no CPU fallback)", file=sys.stderr)
        return 2
    dev = torch.device("cuda:0")
    vals, offsets, _ = synth.config3_device(seed=1002, n_refs=args.refs, n_sample=1000, device="cuda:0")
    n = args.refs
    torch.cuda.synchronize()
    db = RefDB.from_device(vals.data_ptr(), offsets.data_ptr(), n, flags=4)
    stream = torch.cuda.Stream()
    db.set_stream(stream.cuda_stream)
    samples = make_samples(torch, synth, vals, offsets, n, args.batch, args.sample_hashes, 77, dev)
    soff = torch.zeros(args.batch + 1, dtype=torch.int64, device=dev)
    soff[1:] = torch.cumsum(torch.tensor([s.numel() for s in samples], device=dev), 0)
    cat = torch.cat(samples).contiguous()
    total = int(cat.numel())
    out = torch.zeros(3, args.batch, n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        db.run_batch_device(cat.data_ptr(), soff.data_ptr(), args.batch, total, out[0].data_ptr(), out[1].data_ptr(),
                            out[2].data_ptr())

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            step()
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    tm = db.timing()

    # parity of some rows against the one-sample tile path (7 launches) on the same handle
    single = torch.zeros(3, n, dtype=torch.int32, device=dev)
    ok = True
    t_single = []
    for s in sorted(set([0, 1, args.batch // 2, args.batch - 1])):
        smp = samples[s].contiguous()
        with torch.cuda.stream(stream):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            db.run_device(smp.data_ptr(), smp.numel(), single[0].data_ptr(), single[1].data_ptr(), single[2].data_ptr())
            b.record(stream)
        stream.synchronize()
        t_single.append(a.elapsed_time(b))
        ok = ok and bool(torch.equal(single, out[:, s, :]))
    res = {
        "metric": "batched yacht run: samples/sec against one resident database",
        "value": round(args.batch / (ms / 1e3), 1),
        "unit": "samples/s",
        "queries_per_s": round(args.batch * n / (ms / 1e3), 1),
        "ms_per_batch": round(ms, 4),
        "ms_per_sample_in_batch": round(ms / args.batch, 5),
        "ms_one_sample_tile_path": round(float(np.median(t_single)), 4),
        "config": {"workload": f"{args.batch} samples x ~{args.sample_hashes} hashes vs {n} references "
                               f"({int(offsets[-1])} hashes), directory-indexed", "total_sample_hashes": total},
        "kernel_ms": {"lookup": round(float(tm["ms_overlap_kernel"]), 4), "exclusive": round(float(tm["ms_exclusive_kernels"]), 4)},
        "rows_equal_one_sample_path": ok,
    }
    print(json.dumps(res), flush=True)
    db.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
