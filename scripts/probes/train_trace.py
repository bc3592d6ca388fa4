"""Host-side phase times of yh_db_create + yh_pairwise at configs[3] (YH_DEBUG_TUNING=1 YH_TRACE_BUILD=1 prints the phases of the
chunked upload to stderr)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from yacht_amd import synth
from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY

values, offsets = synth.config4(seed=1003, n_clusters=2000, size=5000)
for it in range(3):
    print(f"--- pass {it}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    db = RefDB(values, offsets, flags=YH_DB_PAIRWISE_ONLY)
    t1 = time.perf_counter()
    p = db.pairwise(0.95 ** 31)
    t2 = time.perf_counter()
    tm = db.timing()
    db.close()
    t3 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.3f} ms  pairwise {1e3*(t2-t1):.3f} ms  close {1e3*(t3-t2):.3f} ms  build kernels {tm['ms_db_build']:.3f}  pair kernels {tm['ms_pairwise_kernels']:.3f}", file=sys.stderr, flush=True)
