#!/usr/bin/env python3
"""GPU box: what the hits cost k_index_lookup.  Same database (configs[2] scale), three 1e6-hash samples:
the bench sample (200 genomes present), pure noise (no hash of the database), only database hashes; and the
83 k-hash real-hit-shape sample.  YH_INDEX_TILE=0|1|2|4 forces the kernel form."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yacht_amd import _lib, synth  # noqa: E402
from yacht_amd.engine import RefDB  # noqa: E402

values, offsets, sample = synth.config3_device(seed=1002, n_refs=85_205, n_sample=1_000_000, device="cuda:0")
n = offsets.numel() - 1
mh = synth.max_hash_for_scaled(1000)
g = torch.Generator(device="cuda:0")
g.manual_seed(5)
ROT = 8  # distinct samples per kind, rotated: 8 x 64 MB of buckets do not stay in the 256 MB Infinity Cache
kinds = {"bench": [sample], "noise": [], "allhit": [], "real": []}
for i in range(ROT):
    if i:
        kinds["bench"].append(synth.sample_device(values, offsets, seed=2000 + i, n_sample=1_000_000, n_present=200))
    noise = torch.unique(torch.randint(0, mh, (1_050_000,), generator=g, device="cuda:0", dtype=torch.int64))
    kinds["noise"].append(noise[~torch.isin(noise, values)][:1_000_000].contiguous())
    pick = torch.randperm(values.numel(), generator=g, device="cuda:0")[:1_100_000]
    kinds["allhit"].append(torch.unique(values[pick])[:1_000_000].contiguous())
    kinds["real"].append(synth.sample_device(values, offsets, seed=77 + i, n_sample=83_000, shape="real"))
sv = torch.sort(values).values
shared = torch.unique(sv[1:][sv[1:] == sv[:-1]])
del sv
# the bench samples without their database-shared hashes (no posting-list walk), topped up with noise to the same size
kinds["bench_noshared"] = []
for i, s_ in enumerate(kinds["bench"]):
    keep = s_[~torch.isin(s_, shared)]
    fill = kinds["noise"][i][: s_.numel() - keep.numel()]
    kinds["bench_noshared"].append(torch.unique(torch.cat([keep, fill])).contiguous())
# the real-shape samples: without their shared hashes, and noise of the same size
kinds["real_noshared"] = [s_[~torch.isin(s_, shared)].contiguous() for s_ in kinds["real"]]
kinds["noise_83k"] = [x[:: 12][:83_000].contiguous() for x in kinds["noise"]]
# isolate-like samples: two whole genomes (every hash hits one of two counters) + a little noise
sizes_all = (offsets[1:] - offsets[:-1])
big = torch.argsort(sizes_all, descending=True)[:64].tolist()
kinds["isolate"] = []
for i in range(ROT):
    a_, b_ = big[2 * i], big[2 * i + 1]
    parts = [values[int(offsets[a_]):int(offsets[a_ + 1])], values[int(offsets[b_]):int(offsets[b_ + 1])], kinds["noise"][i][:2000]]
    kinds["isolate"].append(torch.unique(torch.cat(parts)).contiguous())
del shared
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n)
out = torch.zeros(3, n, dtype=torch.int32, device="cuda:0")
res = {}


def step(s):
    db.run_device(s.data_ptr(), s.numel(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())


for mode, mname in ((_lib.YH_LOOKUP_INDEXED, "indexed"),):
    db.set_lookup(mode)
    for name, ss in kinds.items():
        for s in ss:
            step(s)
        db.synchronize()
        db.timing()
        for i in range(64):
            step(ss[i % ROT])
        db.synchronize()
        tm = db.timing()
        res[f"{mname}_{name}"] = {"hashes": int(ss[-1].numel()), "hits": int(out[0].long().sum()),
                                  "refs_hit": int((out[0] > 0).sum()),
                                  "lookup_us": round(1e3 * float(tm["ms_overlap_kernel"]), 2),
                                  "excl_us": round(1e3 * float(tm["ms_exclusive_kernels"]), 2)}
print(os.environ.get("YH_INDEX_TILE", "default"), json.dumps(res))
