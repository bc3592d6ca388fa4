import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from yacht_amd import synth, _lib
from yacht_amd.engine import RefDB
stage = sys.argv[1]
values, offsets, _ = synth.config3_like(seed=5, n_refs=4000, n_sample=1000, n_present=5)
n = offsets.size - 1
refs = [values[int(offsets[j]):int(offsets[j + 1])] for j in range(n)]
rng = np.random.default_rng(1)
big = synth.sample_from_refs(rng, refs, rng.choice(n, 40, replace=False), 0.5, 600_000)
small = synth.sample_from_refs(rng, refs, rng.choice(n, 10, replace=False), 0.5, 30_000)
print("data ready", flush=True)
db = RefDB(values, offsets)
print("db ready", flush=True)
if stage == "plain":
    for s in (small, big):
        ov, e, m = db.run_counts(s); print("run_counts ok", s.size, int(ov.sum()), flush=True)
if stage == "fused":
    d = torch.from_numpy(big.view(np.int64).copy()).cuda()
    bufs = [torch.zeros(3, n, dtype=torch.int32, device="cuda") for _ in range(3)]
    for i in range(int(sys.argv[2])):
        b = bufs[i % 3]
        db.run_device_pipelined(d.data_ptr(), d.numel(), b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
        torch.cuda.synchronize(); print("launch", i, "done", flush=True)
    db.run_device_join(); torch.cuda.synchronize(); print("join done", int(bufs[0][0].sum()), flush=True)
