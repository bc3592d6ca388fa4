"""What does the batched kernel cost for ONE sample (and 2, 4, 8, 32)?  If k_batch_lookup -- one hash per lane, 256-lane
workgroups, chunks of all samples interleaved, one global atomic per hit -- runs a single 1e6-hash sample near its
21 us per 1e6 hashes, a single-sample lookup of that shape would beat the 32 us of the tiled one."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from yacht_amd import synth
from yacht_amd.engine import RefDB

dev = torch.device("cuda:0")
gen = dict(cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
plan = synth.global_db_plan(1, 85205, **gen)
values, offsets = synth.global_db_refs_device(plan, np.arange(0, 85205), device=str(dev))
n = 85205
samples = [synth.global_db_sample_device(plan, 1001 + i, n_sample=1_000_000, n_present=200, device=str(dev)) for i in range(64)]
db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n, device=0)
st = torch.cuda.Stream()
db.set_stream(st.cuda_stream)
for B in (1, 2, 4, 8, 16, 32, 64):
    reps = 64 // B
    packs = []
    for g in range(reps):
        ss = samples[g * B:(g + 1) * B]
        cat = torch.cat(ss).contiguous()
        off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(torch.tensor([s.numel() for s in ss], device=dev), 0)
        packs.append((cat, off, int(cat.numel())))
    outs = [torch.zeros(B, n, dtype=torch.int32, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    def go(k):
        for i in range(k):
            cat, off, tot = packs[i % reps]
            db.run_batch_device(cat.data_ptr(), off.data_ptr(), B, tot, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())
        db.synchronize()
    go(reps)
    t0 = time.perf_counter(); go(4 * reps); dt = time.perf_counter() - t0
    tm = db.timing()
    print(f"B={B:2d}: {dt / (4 * reps * B) * 1e6:7.2f} us per sample   lookup kernel {tm['ms_overlap_kernel'] * 1e3 / B:7.2f} us per sample", flush=True)
