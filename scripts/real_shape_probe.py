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
samples = [synth.global_db_sample_device(plan, 6002 + i, n_sample=83_000 if shape == "real" else 1_000_000, device="cuda:0",
                                         shape=shape) for i in range(4)]
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), 85_205)
c = torch.zeros((3, 85_205), device="cuda:0", dtype=torch.int32)
torch.cuda.synchronize()
for i in range(steps):
    s = samples[i % 4]
    db.run_device(s.data_ptr(), s.numel(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())
db.synchronize()
print("overlapping", int((c[0] != 0).sum()), db.timing())
