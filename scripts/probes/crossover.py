#!/usr/bin/env python3
"""GPU box: where the streaming lookup overtakes the sample-driven one (yh_api.hip, prefer_indexed).
rs214-scale database, samples of 1e6 .. 3.2e7 hashes (200 genomes present + noise), both lookups forced."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yacht_amd import _lib, synth  # noqa: E402
from yacht_amd.engine import RefDB  # noqa: E402

values, offsets, _ = synth.config3_device(seed=1002, n_refs=85_205, n_sample=1_000_000, device="cuda:0")
n = offsets.numel() - 1
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n)
out = torch.zeros(3, n, dtype=torch.int32, device="cuda:0")
res = {}
for ns in (1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000, 32_000_000):
    ss = [synth.sample_device(values, offsets, seed=900 + i, n_sample=ns, n_present=200) for i in range(3)]
    row = {"auto": "indexed" if db.lookup_choice(ns) == _lib.YH_LOOKUP_INDEXED else "stream"}
    for mode, name in ((_lib.YH_LOOKUP_INDEXED, "indexed"), (_lib.YH_LOOKUP_STREAM, "stream")):
        db.set_lookup(mode)
        for s in ss:
            db.run_device(s.data_ptr(), s.numel(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
        db.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for i in range(12):
            s = ss[i % 3]
            db.run_device(s.data_ptr(), s.numel(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
        db.synchronize()
        row[name + "_step_us"] = round((time.perf_counter() - t0) / 12 * 1e6, 1)
    db.set_lookup(_lib.YH_LOOKUP_AUTO)
    res[ns] = row
    del ss
print(json.dumps(res))
