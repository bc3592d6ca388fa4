"""Where yh_db_create's time goes at rs214 scale, device input, inside a torch process (what bench.py does) -- run with
YH_DEBUG_TUNING=1 YH_TRACE_BUILD=1: the phases of the build go to stderr; this prints the wall clock and ms_db_build
of three creates in a row (the first pays the first-use allocations of the buffer cache)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from yacht_amd import synth
from yacht_amd.engine import RefDB, YH_DB_DEFAULT

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 85205
dev = "cuda:0"
plan = synth.global_db_plan(1002, n_refs, cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
values, offsets = synth.global_db_refs_device(plan, np.arange(0, n_refs), device=dev)
torch.cuda.synchronize()
print(f"{n_refs} refs, {values.numel()} hashes", file=sys.stderr, flush=True)
for it in range(6):
    print(f"--- create {it}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_refs, device=0, flags=YH_DB_DEFAULT)
    t1 = time.perf_counter()
    tm = db.timing()
    inf = db.info()
    db.close()
    t2 = time.perf_counter()
    print(f"create {1e3 * (t1 - t0):.1f} ms wall, ms_db_build {tm['ms_db_build']:.1f}, close {1e3 * (t2 - t1):.1f} ms, "
          f"device_bytes {inf['device_bytes'] / 1e9:.2f} GB", file=sys.stderr, flush=True)
