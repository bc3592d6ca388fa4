#!/usr/bin/env python3
"""The row pass (k_pair_rows) at the scale the reference publishes: 85 205 sketches, 3.3e8 hashes, a train handle made once, yh_pairwise
timed three times (thresholds a hair apart: the handle caches a call's result).  The tuning environment decides lanes per row, 16- or
32-bit counts and columns per block: scripts/probes/rs214_rows_sweep.sh."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import YH_DB_PAIRWISE_ONLY, RefDB  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 85_205
values, offsets, _ = synth.config3_device(seed=1002, n_refs=n_refs, n_sample=1000, device="cuda:0")
torch.cuda.synchronize()
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_refs, flags=YH_DB_PAIRWISE_ONLY)
c = 0.95 ** 31
out = []
digest = ""
for k in range(4):
    t0 = time.perf_counter()
    pi, pj, pc = db.pairwise(c * (1.0 - 1e-13 * k))
    t1 = time.perf_counter()
    out.append((1e3 * (t1 - t0), float(db.timing()["ms_pairwise_kernels"]), int(pi.size)))
    if k == 0:  # (the digest scripts/train_rs214_parity.py prints for the pairs: equal to the oracle port's at N = 85 205)
        import hashlib

        import numpy as np

        h = hashlib.sha256()
        for a in (pi, pj, pc):
            h.update(np.ascontiguousarray(a).tobytes())
        digest = h.hexdigest()[:32]
db.close()
print("%-60s call ms %s  kernels ms %s  pairs %d  digest %s" % (" ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("YH_PAIR")) or "(default)",
                                                     " ".join("%.2f" % x[0] for x in out[1:]), " ".join("%.2f" % x[1] for x in out[1:]), out[-1][2], digest), flush=True)
