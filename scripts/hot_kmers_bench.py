#!/usr/bin/env python3
"""VERDICT r04 "next" 5, the bench leg: databases with a few HOT k-mers -- 50 hashes held by 5 000 ... 40 000 references each
(conserved rRNA 31-mers across GTDB are that) -- against the same databases without them.
  train  configs[3] (10 000 sketches of ~5 000 hashes) + 50 hashes in 5 000 - 10 000 sketches each: the train handle
         (yh_db_create_device, YH_DB_PAIRWISE_ONLY) uniform vs hot; pairs / statistics / selection against the oracle port
  run    rs214 scale (85 205 references, 3.3e8 hashes) + 50 hashes in 5 000 - 40 000 references each: the full handle uniform
         vs hot; the counts of a sample that holds half of the hot hashes against the oracle
Until round 4 ONE overflowing bucket sent the whole database to rocPRIM's radix sort (YH_NO_SPILL=1 / YH_NO_PIECES*=1 behind
the tuning gate still do: `--old` times that too).   usage (GPU box): python scripts/hot_kmers_bench.py > gpurun_out/hot_kmers.json"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import YH_DB_DEFAULT, YH_DB_PAIRWISE_ONLY, RefDB, train_select  # noqa: E402

DEV = "cuda:0"


def inject(values, offsets, n_hot, lo, hi, seed):
    """values / offsets (int64 tensors on the device) + n_hot new hashes, each put into lo .. hi randomly chosen references."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = int(offsets.numel() - 1)
    mh = synth.max_hash_for_scaled(1000)
    hot = torch.randint(1, mh, (n_hot,), generator=g, dtype=torch.int64)
    assert not bool(torch.isin(hot.to(DEV), values).any())
    sizes = offsets[1:] - offsets[:-1]
    ref_of = torch.repeat_interleave(torch.arange(n, device=DEV, dtype=torch.int64), sizes)
    add_v, add_r, holders = [], [], []
    for h in hot.tolist():
        m = int(torch.randint(lo, hi + 1, (1,), generator=g).item())
        who = torch.randperm(n, generator=g)[:m]
        add_v.append(torch.full((m,), h, dtype=torch.int64))
        add_r.append(who)
        holders.append(m)
    all_v = torch.cat([values, torch.cat(add_v).to(DEV)])
    all_r = torch.cat([ref_of, torch.cat(add_r).to(DEV)])
    del ref_of
    i1 = torch.argsort(all_v, stable=True)  # (hashes are < 2^63: signed order is unsigned order)
    v1, r1 = all_v[i1], all_r[i1]
    del all_v, all_r, i1
    i2 = torch.argsort(r1, stable=True)
    out_v = v1[i2].contiguous()
    cnt = torch.bincount(r1, minlength=n)
    out_o = torch.zeros(n + 1, dtype=torch.int64, device=DEV)
    out_o[1:] = torch.cumsum(cnt, 0)
    return out_v, out_o, hot, holders


def time_create(values, offsets, n, flags, reps=5):
    from yacht_amd import _lib

    _lib.pool_release()  # (every series starts with an empty buffer cache: the first create of each is the warm-up)
    ts, info = [], None
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n, flags=flags)
        db.synchronize()
        ts.append(time.perf_counter() - t0)
        info = db.info()
        bms = db.timing()["ms_db_build"]
        db.close()
    return float(np.median(ts[1:])) * 1e3, float(bms), dict(info, ms_in_hipMalloc_of_the_series=_lib.alloc_stats()["ms_in_driver"])


def main() -> int:
    from oracle import oracle

    out = {}
    threads = oracle.hardware_threads()
    # ---- train: configs[3]
    v, o = synth.config4(seed=1003, n_clusters=2000, size=5000)
    vt = torch.from_numpy(v.view(np.int64)).to(DEV)
    ot = torch.from_numpy(o.astype(np.int64)).to(DEV)
    n = o.size - 1
    hv, ho, hot, holders = inject(vt, ot, 50, 5000, 10000, 11)
    uni = time_create(vt, ot, n, YH_DB_PAIRWISE_ONLY)
    hotc = time_create(hv, ho, n, YH_DB_PAIRWISE_ONLY)
    c = 0.95 ** 31
    h_values = hv.cpu().numpy().view(np.uint64)
    h_offsets = ho.cpu().numpy().astype(np.uint64)
    sizes = np.diff(h_offsets).astype(np.uint32)
    with RefDB.from_device(hv.data_ptr(), ho.data_ptr(), n, flags=YH_DB_PAIRWISE_ONLY) as db:
        t0 = time.perf_counter()
        pi, pj, pc = db.pairwise(c)
        t_pair = time.perf_counter() - t0
        stats = tuple(int(x) for x in db.index_stats())
    sel = train_select(sizes, pi, pj)
    t0 = time.perf_counter()
    wi, wj, wc, wstats = oracle.train_pairs(h_values, h_offsets, c, threads=threads)
    wsel = oracle.train_select(sizes, wi, wj)
    t_or = time.perf_counter() - t0
    out["train_configs3"] = {
        "hot_hashes": 50, "holders_min_max": [min(holders), max(holders)], "pairs_of_hot_hashes": int(sum(holders)),
        "create_ms_uniform": round(uni[0], 3), "create_ms_hot": round(hotc[0], 3), "ratio": round(hotc[0] / uni[0], 3),
        "build_kernels_ms_uniform": round(uni[1], 3), "build_kernels_ms_hot": round(hotc[1], 3),
        "sort_path_hot": hotc[2]["sort_path"], "n_spilled_buckets": hotc[2]["n_spilled_buckets"], "n_spilled_pairs": hotc[2]["n_spilled_pairs"],
        "pairwise_s_hot": round(t_pair, 4), "oracle_s": round(t_or, 1),
        "pairs_equal": bool(np.array_equal(pi, wi) and np.array_equal(pj, wj) and np.array_equal(pc, wc)),
        "stats_equal": stats == tuple(int(x) for x in wstats), "selection_equal": bool(np.array_equal(sel, wsel)), "pairs_kept": int(pi.size)}
    del vt, ot, hv, ho
    torch.cuda.empty_cache()
    # ---- run: rs214 scale
    n = 85_205
    plan = synth.global_db_plan(1002, n, cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
    vt, ot = synth.global_db_refs_device(plan, np.arange(n), device=DEV)
    hv, ho, hot, holders = inject(vt, ot, 50, 5000, 40000, 12)
    uni = time_create(vt, ot, n, YH_DB_DEFAULT, reps=3)
    hotc = time_create(hv, ho, n, YH_DB_DEFAULT, reps=3)
    sample = synth.global_db_sample_device(plan, 2002, n_sample=1_000_000, n_present=200, device=DEV)
    sample = torch.unique(torch.cat([sample, hot[:25].to(DEV)]))  # (sorted; half of the hot hashes are in the sample)
    h_values = hv.cpu().numpy().view(np.uint64)
    h_offsets = ho.cpu().numpy().astype(np.uint64)
    h_sample = sample.cpu().numpy().view(np.uint64)
    with RefDB.from_device(hv.data_ptr(), ho.data_ptr(), n, flags=YH_DB_DEFAULT) as db:
        cnt = torch.zeros((3, n), dtype=torch.int32, device=DEV)
        torch.cuda.synchronize()
        db.run_device(sample.data_ptr(), sample.numel(), cnt[0].data_ptr(), cnt[1].data_ptr(), cnt[2].data_ptr())
        db.synchronize()
        got = cnt.cpu().numpy().view(np.uint32)
    t0 = time.perf_counter()
    w_ov = oracle.overlap(h_values, h_offsets, h_sample, threads=threads)
    w_e, w_m = oracle.exclusive(h_values, h_offsets, w_ov > 0, h_sample)
    t_or = time.perf_counter() - t0
    out["run_rs214"] = {
        "hot_hashes": 50, "holders_min_max": [min(holders), max(holders)], "pairs_of_hot_hashes": int(sum(holders)),
        "create_ms_uniform": round(uni[0], 2), "create_ms_hot": round(hotc[0], 2), "ratio": round(hotc[0] / uni[0], 3),
        "build_kernels_ms_uniform": round(uni[1], 2), "build_kernels_ms_hot": round(hotc[1], 2),
        "sort_path_uniform": uni[2]["sort_path"], "sort_path_hot": hotc[2]["sort_path"],
        "n_spilled_buckets": hotc[2]["n_spilled_buckets"], "n_spilled_pairs": hotc[2]["n_spilled_pairs"],
        "refs_overlapping": int((w_ov > 0).sum()), "oracle_s": round(t_or, 1),
        "counts_equal": bool(np.array_equal(got[0], w_ov) and np.array_equal(got[1], w_e) and np.array_equal(got[2], w_m))}
    if "--old" in sys.argv:  # round 4's behaviour on the hot databases, in child processes (the switches are read once per process)
        pass
    print(json.dumps(out), flush=True)
    ok = out["train_configs3"]["pairs_equal"] and out["train_configs3"]["stats_equal"] and out["train_configs3"]["selection_equal"] and out["run_rs214"]["counts_equal"]
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
