#!/usr/bin/env python3
"""db_build_ms of the FIRST handle of a process, over several fresh processes (VERDICT r04 "next" 7: the driver's run of
bench.py saw 601 ms where the builder's sessions see 57-60; profiles/r04/malloc_probe.txt: a hipMalloc behind frees now
and then takes seconds on this pool).  Each child generates the rs214-scale database in HBM with torch (as bench.py does),
then creates ONE handle and reports the build's device time, its wall time and the host time spent inside hipMalloc
(yh_alloc_stats) -- under `--mode`:
    plain      as bench.py does it
    settle     torch.cuda.empty_cache() + synchronize + 50 ms of sleep in front of the create (are torch's frees what the
               driver is still busy with?)
    prewarmN   one N-GiB block allocated and handed back to the driver before anything else (default 48): if the stall is a
               once-per-process cost of the first big allocation, it moves there -- where a helper thread could pay it while
               the command still parses its inputs
usage (GPU box):  python scripts/probes/fresh_build_probe.py [--runs 6]  -> one line per child + p50 / p90 per mode"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from yacht_amd import synth, _lib
from yacht_amd.engine import RefDB, YH_DB_DEFAULT
mode = sys.argv[1]
t_start = time.perf_counter()
plan = synth.global_db_plan(1002, 85205, cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
values, offsets = synth.global_db_refs_device(plan, np.arange(85205), device="cuda:0")
torch.cuda.synchronize()
if mode == "settle":
    torch.cuda.empty_cache(); torch.cuda.synchronize(); time.sleep(0.05)
t_pre = 0.0
if mode.startswith("prewarm"):  # one big allocation made and given back to the driver first: is the cost once per process?
    gb = int(mode[len("prewarm"):] or 48)
    t0 = time.perf_counter()
    blk = torch.empty(gb << 30, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    del blk
    torch.cuda.empty_cache(); torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
m0 = _lib.alloc_stats()
t0 = time.perf_counter()
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), 85205, flags=YH_DB_DEFAULT)
db.synchronize()
t1 = time.perf_counter()
m1 = _lib.alloc_stats()
print(json.dumps({"mode": mode, "db_build_ms": round(float(db.timing()["ms_db_build"]), 2), "create_wall_ms": round(1e3 * (t1 - t0), 2),
                  "driver_allocs": m1["driver_allocs"] - m0["driver_allocs"], "ms_in_hipMalloc": round(m1["ms_in_driver"] - m0["ms_in_driver"], 1),
                  "process_s_before_create": round(t0 - t_start, 2), "prewarm_s": round(t_pre, 3)}))
db.close()
''' % ROOT


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--modes", default="plain,settle")
    args = ap.parse_args()
    import numpy as np

    for mode in args.modes.split(","):
        rows = []
        for _ in range(args.runs):
            p = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True, timeout=600)
            ln = [x for x in p.stdout.splitlines() if x.startswith("{")]
            if not ln:
                print(mode, "child failed:", p.stderr[-300:])
                continue
            rows.append(json.loads(ln[-1]))
            print(ln[-1], flush=True)
        if rows:
            b = [r["db_build_ms"] for r in rows]
            w = [r["create_wall_ms"] for r in rows]
            print(json.dumps({"mode": mode, "runs": len(rows), "db_build_ms_p50": round(float(np.percentile(b, 50)), 1),
                              "db_build_ms_p90": round(float(np.percentile(b, 90)), 1), "db_build_ms_max": max(b),
                              "create_wall_ms_p50": round(float(np.percentile(w, 50)), 1), "create_wall_ms_p90": round(float(np.percentile(w, 90)), 1),
                              "ms_in_hipMalloc_max": max(r["ms_in_hipMalloc"] for r in rows)}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
