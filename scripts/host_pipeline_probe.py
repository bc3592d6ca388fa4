#!/usr/bin/env python3
"""The host-inclusive loop alone (yh_run_submit / yh_run_wait, pinned buffers) for tracing:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -- python3 scripts/host_pipeline_probe.py [steps] [depth]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import PinnedArray, RefDB  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_refs = int(sys.argv[3]) if len(sys.argv) > 3 else 85_205
plan = synth.global_db_plan(1002, n_refs)
values, offsets = synth.global_db_refs_device(plan, np.arange(n_refs), device="cuda:0")
K = 4
samples = []
for i in range(K):
    s = synth.global_db_sample_device(plan, 2002 + i, n_sample=1_000_000, device="cuda:0")
    pa = PinnedArray(int(s.numel()), np.uint64)
    pa.array[:] = s.cpu().numpy().view(np.uint64)
    samples.append(pa)
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_refs)
blk = [PinnedArray(3 * n_refs, np.uint32) for _ in range(depth)]
outs = [[b.array[k * n_refs:(k + 1) * n_refs] for k in range(3)] for b in blk]
torch.cuda.synchronize()
done = []
t_sub, t_wait = [], []
t0 = time.perf_counter()
for i in range(steps + depth):
    slot = i % depth
    if i >= depth:
        ta = time.perf_counter()
        db.run_wait(slot)
        done.append(time.perf_counter())
        t_wait.append(done[-1] - ta)
    if i < steps:
        o = outs[slot]
        ta = time.perf_counter()
        db.run_submit(slot, samples[i % K].array, o[0], o[1], o[2])
        t_sub.append(time.perf_counter() - ta)
gaps = np.diff(np.asarray([t0] + done)) * 1e3
big = np.flatnonzero(gaps > 2.0)
print("depth", depth, "ms/step", round(float((done[-1] - t0) / steps * 1e3), 4), "median gap", round(float(np.median(gaps)), 4),
      "stalls >2ms at", big.tolist(), np.round(gaps[big], 1).tolist())
ts, tw = np.asarray(t_sub) * 1e3, np.asarray(t_wait) * 1e3
print("  submit ms: median", round(float(np.median(ts)), 4), "max", round(float(ts.max()), 2), "at", int(ts.argmax()),
      "| wait ms: median", round(float(np.median(tw)), 4), "max", round(float(tw.max()), 2), "at", int(tw.argmax()))
print("  big submits", np.flatnonzero(ts > 2).tolist(), "big waits", np.flatnonzero(tw > 2).tolist())
