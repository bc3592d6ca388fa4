#!/usr/bin/env python3
"""The real-hit-shape step alone (bench.py's `real_shape` leg) for profilers:
   rocprofv3 --pmc ... -- python3 scripts/real_shape_probe.py [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import RefDB  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shape = sys.argv[2] if len(sys.argv) > 2 else "real"
plan = synth.global_db_plan(1002, 85_205)
values, offsets = synth.global_db_refs_device(plan, np.arange(85_205), device="cuda:0")
if shape == "strat":  # one hash per equal-width stratum of the hash range (what scripts/probes/gather_probe.hip looks up)
    mh = synth.max_hash_for_scaled(1000)
    n = 1_000_000
    step = mh // n
    g = torch.Generator(device="cuda:0")
    samples = []
    for i in range(4):
        g.manual_seed(i)
        samples.append((torch.arange(n, device="cuda:0", dtype=torch.int64) * step
                        + torch.randint(0, step, (n,), generator=g, device="cuda:0", dtype=torch.int64)).contiguous())
else:
    samples = [synth.global_db_sample_device(plan, 6002 + i, n_sample=83_000 if shape == "real" else 1_000_000, device="cuda:0",
                                             shape=shape, n_present=int(os.environ.get("PRESENT", "200"))) for i in range(4)]
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), 85_205)
c = torch.zeros((3, 85_205), device="cuda:0", dtype=torch.int32)
torch.cuda.synchronize()
for i in range(steps):
    s = samples[i % 4]
    db.run_device(s.data_ptr(), s.numel(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())
db.synchronize()
t = db.timing()
print(shape, "overlapping", int((c[0] != 0).sum()), "lookup kernel ms", round(t["ms_overlap_kernel"], 4), "excl ms", round(t["ms_exclusive_kernels"], 4))
