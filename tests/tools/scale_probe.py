#!/usr/bin/env python3
"""One GPU's shard of BASELINE.json configs[4] (GTDB full, ~400 k genomes at scaled=100, sharded
over 8 GPUs): ~50 000 references x ~39 000 hashes = ~2 x 10^9 reference hashes (> 2^31 positions in
every array) against a 10^7-hash sample.  Not a bench line: a maximum-size parity case.

    python tests/tools/scale_probe.py [--refs 50000] [--median 33000] [--sample 10000000] [--oracle auto|yes|no]

Checks, all bit-exact:
  * overlap from the streaming kernel == overlap from k_overlap_bsearch (independent kernel over
    the plain CSR) == a torch searchsorted count of the same thing;
  * overlap / exclusive counts == the CPU oracle, when the host has the memory for it (--oracle).
Prints one JSON line with sizes, timings and the verdicts.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def host_free_gib() -> float:
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=56_000)
    ap.add_argument("--median", type=float, default=33_000.0)
    ap.add_argument("--sample", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--oracle", choices=("auto", "yes", "no"), default="auto")
    args = ap.parse_args(argv)

    import torch

    from yacht_amd import _lib, synth
    from yacht_amd.engine import RefDB

    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    # torch.sort takes at most 2^31-1 elements: generate the references in groups and concatenate
    groups = max(1, int(args.refs * args.median * 1.2 / 1.5e9) + 1)
    per = (args.refs + groups - 1) // groups
    vs, os_, ss, base = [], [], [], 0
    for gi in range(groups):
        k = min(per, args.refs - gi * per)
        if k <= 0:
            break
        v, o, s = synth.config3_device(seed=4004 + gi, n_refs=k, n_sample=args.sample // groups, device="cuda:0",
                                       median=args.median, sigma=0.6, lo=3000, hi=150_000, scaled=100,
                                       n_present=max(200 // groups, 1))
        vs.append(v)
        os_.append((o[1:] if gi else o) + base)
        base += int(v.numel())
        ss.append(s)
    vals = torch.cat(vs)
    del vs
    offsets = torch.cat(os_).contiguous()
    sample = torch.unique(torch.cat(ss)).contiguous()
    del ss
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    n, H = args.refs, int(vals.numel())
    torch.cuda.empty_cache()  # the generator's sort buffers go back to the driver: the library allocates with hipMalloc
    t0 = time.perf_counter()
    db = RefDB.from_device(vals.data_ptr(), offsets.data_ptr(), n, flags=2)  # YH_DB_KEEP_CSR
    t_build = time.perf_counter() - t0
    info = db.info()
    stream = torch.cuda.Stream()
    db.set_stream(stream.cuda_stream)
    out = torch.zeros(3, n, dtype=torch.int32, device=dev)
    chk = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        db.run_device(sample.data_ptr(), sample.numel(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())

    default_choice = db.lookup_choice(int(sample.numel()))
    # the sample-driven lookup first (one bucket read per sample hash), then the streaming kernel: same counts
    db.set_lookup(_lib.YH_LOOKUP_INDEXED)
    step()
    stream.synchronize()
    db.timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    stream.synchronize()
    ms_step_indexed = (time.perf_counter() - t0) / args.steps * 1e3
    tm_idx = db.timing()
    out_indexed = out.clone()
    db.set_lookup(_lib.YH_LOOKUP_STREAM)
    step()
    stream.synchronize()
    db.timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    stream.synchronize()
    ms_step = (time.perf_counter() - t0) / args.steps * 1e3
    tm = db.timing()
    same_paths = bool(torch.equal(out, out_indexed))
    db.set_lookup(_lib.YH_LOOKUP_AUTO)

    db.overlap_bsearch_device(sample.data_ptr(), sample.numel(), chk.data_ptr())
    stream.synchronize()
    same_bsearch = bool(torch.equal(chk, out[0]))

    # torch count of reference hashes that are in the sample, in slices (independent of both kernels)
    total = 0
    for a in range(0, H, 1 << 27):
        v = vals[a:a + (1 << 27)]
        idx = torch.searchsorted(sample, v).clamp_(max=sample.numel() - 1)
        total += int((sample[idx] == v).sum())
    same_total = total == int(out[0].long().sum())

    oracle_ok = None
    need_gib = H * 8 / (1 << 30) * 1.6 + 8
    if args.oracle == "yes" or (args.oracle == "auto" and host_free_gib() > need_gib + 16):
        from oracle import oracle

        hv = vals.cpu().numpy().view(np.uint64)
        ho = offsets.cpu().numpy().view(np.uint64)
        hs = sample.cpu().numpy().view(np.uint64)
        w_ov = oracle.overlap(hv, ho, hs, threads=oracle.hardware_threads())
        w_e, w_m = oracle.exclusive(hv, ho, w_ov > 0, hs)
        got = out.cpu().numpy().view(np.uint32)
        oracle_ok = bool(np.array_equal(got[0], w_ov) and np.array_equal(got[1], w_e) and np.array_equal(got[2], w_m))

    key_bytes = int(info.get("stream_bytes", 3 * H)) + 8 * sample.numel()  # what the streaming kernel has to read
    k_ms = float(tm["ms_overlap_kernel"])
    res = {
        "workload": f"one GPU's shard of configs[4]: {n} references, {H} hashes (scaled=100), sample {sample.numel()} hashes",
        "positions_exceed_2^31": H > (1 << 31),
        "stream_layout": info.get("stream_layout"), "stream_bytes": info.get("stream_bytes"), "db_hbm_bytes": info["device_bytes"],
        "seconds": {"generate": round(t_gen, 2), "build": round(t_build, 3)},
        "ms_per_step": round(ms_step, 3), "queries_per_s": round(n / (ms_step / 1e3), 1),
        "ms_per_step_indexed": round(ms_step_indexed, 3), "index_lookup_ms": round(float(tm_idx["ms_overlap_kernel"]), 4),
        "default_lookup": "indexed" if default_choice == _lib.YH_LOOKUP_INDEXED else "stream",
        "indexed_equals_stream": same_paths,
        "k1_ms": round(k_ms, 4), "k1_GBps": round(key_bytes / 1e9 / (k_ms / 1e3), 1) if k_ms else None,
        "overlap_equals_bsearch_kernel": same_bsearch,
        "overlap_sum_equals_torch_count": same_total,
        "equals_cpu_oracle": oracle_ok,
        "host_free_gib": round(host_free_gib(), 1),
    }
    print(json.dumps(res), flush=True)
    db.close()
    return 0 if (same_bsearch and same_total and same_paths and oracle_ok in (None, True)) else 1


if __name__ == "__main__":
    sys.exit(main())
