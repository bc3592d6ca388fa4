#!/usr/bin/env python3
"""GPU box: the batched run (yh_run_batch_device, up to 64 samples per pass) against a loop of single
fused steps (yh_run_device) on the bench workload; both device-resident, rotating samples."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import RefDB  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
values, offsets, s0 = synth.config3_device(seed=1002, n_refs=85_205, n_sample=1_000_000, device="cuda:0")
n = offsets.numel() - 1
shape = sys.argv[2] if len(sys.argv) > 2 else "present"
ns = 1_000_000 if shape == "present" else 83_000  # (isolate: set below)
if shape == "isolate":  # every sample = two whole genomes: all hits of a sample land on two counters
    sizes_all = offsets[1:] - offsets[:-1]
    big = torch.argsort(sizes_all, descending=True)[: 2 * B].tolist()
    samples = [torch.unique(torch.cat([values[int(offsets[big[2 * i]]):int(offsets[big[2 * i] + 1])],
                                       values[int(offsets[big[2 * i + 1]]):int(offsets[big[2 * i + 1] + 1])]])).contiguous()
               for i in range(B)]
    ns = int(samples[0].numel())
else:
    samples = [synth.sample_device(values, offsets, seed=3000 + i, n_sample=ns, n_present=200, shape=shape) for i in range(B)]
cat = torch.cat(samples).contiguous()
soff = torch.zeros(B + 1, dtype=torch.int64, device="cuda:0")
soff[1:] = torch.cumsum(torch.tensor([s.numel() for s in samples], device="cuda:0"), 0)
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n)
single = torch.zeros(B, 3, n, dtype=torch.int32, device="cuda:0")
batch = torch.zeros(3, B, n, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()


def run_single():
    for i, s in enumerate(samples):
        db.run_device(s.data_ptr(), s.numel(), single[i, 0].data_ptr(), single[i, 1].data_ptr(), single[i, 2].data_ptr())


def run_batch():
    db.run_batch_device(cat.data_ptr(), soff.data_ptr(), B, cat.numel(), batch[0].data_ptr(), batch[1].data_ptr(), batch[2].data_ptr())


res = {"samples": B, "shape": shape, "hashes_per_sample": ns}
for name, fn in (("single_loop", run_single), ("batch", run_batch)):
    for _ in range(3):
        fn()
    db.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    db.synchronize()
    res[name + "_ms_per_sample"] = round((time.perf_counter() - t0) / 20 / B * 1e3, 4)
res["equal"] = bool(torch.equal(single.permute(1, 0, 2), batch))
print(json.dumps(res))
