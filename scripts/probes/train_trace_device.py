import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from yacht_amd import synth
from yacht_amd.engine import RefDB, YH_DB_PAIRWISE_ONLY
values, offsets = synth.config4(seed=1003, n_clusters=2000, size=5000)
n = offsets.size - 1
sizes = np.diff(offsets).astype(np.uint32)
dv = torch.from_numpy(values.view(np.int64)).to("cuda:0"); do = torch.from_numpy(offsets.astype(np.int64)).to("cuda:0")
torch.cuda.synchronize()
for it in range(4):
    print(f"--- pass {it}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    db = RefDB.from_device(dv.data_ptr(), do.data_ptr(), n, sizes=sizes, device=0, flags=YH_DB_PAIRWISE_ONLY)
    t1 = time.perf_counter()
    p = db.pairwise(0.95 ** 31)
    t2 = time.perf_counter()
    db.close()
    t3 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.3f} ms  pairwise {1e3*(t2-t1):.3f} ms  close {1e3*(t3-t2):.3f} ms", file=sys.stderr, flush=True)
