"""Do two independent step pipelines on two streams (even / odd samples, a handle each) beat one?  The lookups of consecutive
samples do not depend on each other; on ONE stream the ramp and the tail of every launch are exposed.  bench.py's database
and samples; aggregate microseconds per sample."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from yacht_amd import synth
from yacht_amd.engine import RefDB

dev = torch.device("cuda:0")
gen = dict(cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
plan = synth.global_db_plan(1, 85205, **gen)
values, offsets = synth.global_db_refs_device(plan, np.arange(0, 85205), device=str(dev))
samples = [synth.global_db_sample_device(plan, 1001 + i, n_sample=1_000_000, n_present=200, device=str(dev)) for i in range(8)]
n = 85205
dbs = [RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n, device=0) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for db, st in zip(dbs, streams):
    db.set_stream(st.cuda_stream)
bufs = [[torch.zeros(3, n, dtype=torch.int32, device=dev) for _ in range(3)] for _ in range(2)]
torch.cuda.synchronize()

def run(n_steps, two):
    t0 = time.perf_counter()
    for i in range(n_steps):
        w = (i & 1) if two else 0
        s = samples[i % 8]
        b = bufs[w][(i // (2 if two else 1)) % 3]
        dbs[w].run_device_pipelined(s.data_ptr(), s.numel(), b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
    for db in dbs:
        db.run_device_join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_steps * 1e6

for two in (False, True, False, True):
    run(40, two)
    print("two pipelines" if two else "one pipeline ", " ".join(f"{run(400, two):.2f}" for _ in range(3)), "us per sample", flush=True)
ref = bufs[0][0].clone()
