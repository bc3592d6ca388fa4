cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5; do
YH_DEBUG_TUNING=1 YH_TRACE_BUILD=1 python - <<'PY' 2>&1 | grep -E "yh alloc|RESULT" | head -12
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from yacht_amd import synth, _lib
from yacht_amd.engine import RefDB, YH_DB_DEFAULT
plan = synth.global_db_plan(1002, 85205, cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
values, offsets = synth.global_db_refs_device(plan, np.arange(85205), device="cuda:0")
torch.cuda.synchronize()
free0, tot = torch.cuda.mem_get_info()
t0 = time.perf_counter()
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), 85205, flags=YH_DB_DEFAULT)
db.synchronize()
print("RESULT create_wall_ms %.1f  free before %.1f GB  torch reserved %.1f GB" % (1e3 * (time.perf_counter() - t0), free0 / 2**30, torch.cuda.memory_reserved() / 2**30))
PY
echo ---
done
