"""The distribution sort (yh_sort.hip) on its own: yh_db_create_device(PAIRWISE_ONLY) at configs[3] and at rs214 scale with
YH_DEBUG_TUNING=1 YH_TRACE_BUILD=1 YH_CHECK_SORT=1 -- the sort's own verdict line ([yh sort] ...), the order checked on the
device, and the build's phases."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from yacht_amd import synth
from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY

which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
if which == "cfg3":
    values, offsets = synth.config4(seed=1003, n_clusters=2000, size=5000)
    dv = torch.from_numpy(values.view(np.int64)).cuda()
    do = torch.from_numpy(offsets.astype(np.int64)).cuda()
    n = offsets.size - 1
else:
    n = 85205
    plan = synth.global_db_plan(1002, n, cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
    dv, do = synth.global_db_refs_device(plan, np.arange(n), device="cuda:0")
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    db = RefDB.from_device(dv.data_ptr(), do.data_ptr(), n, flags=YH_DB_PAIRWISE_ONLY)
    t1 = time.perf_counter()
    print(f"create {1e3 * (t1 - t0):.2f} ms, ms_db_build {db.timing()['ms_db_build']:.2f}, stats {db.index_stats()}", file=sys.stderr, flush=True)
    db.close()
