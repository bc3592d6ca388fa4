#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary's host-pointer entry points:
yh_run with the sample in pageable host memory and the three count arrays returned to the host.
Never the bench `value` (that has inputs resident in HBM); printed for the record."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import RefDB  # noqa: E402

values, offsets, sample = synth.config3_device(seed=1002, n_refs=85_205, n_sample=1_000_000, device="cuda:0")
n = offsets.numel() - 1
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n)
h_sample = sample.cpu().numpy().view(np.uint64)
for _ in range(5):
    db.run_counts(h_sample)
t0 = time.perf_counter()
K = 50
for _ in range(K):
    ov, e, m = db.run_counts(h_sample)
dt = (time.perf_counter() - t0) / K
t0 = time.perf_counter()
for _ in range(K):
    ov = db.overlap(h_sample)
dt2 = (time.perf_counter() - t0) / K
print(f"host-buffer yh_run: {dt*1e3:.3f} ms/step = {n/dt/1e6:.1f} M queries/s; "
      f"host-buffer yh_overlap: {dt2*1e3:.3f} ms/step = {n/dt2/1e6:.1f} M queries/s "
      f"(sample {h_sample.nbytes/1e6:.1f} MB H2D, counts {3*4*n/1e6:.2f} MB D2H per step)")
db.close()
