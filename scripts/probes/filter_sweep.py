#!/usr/bin/env python3
"""GPU box: the sample-driven lookup by presence-filter size.  One process per setting (the library reads
YH_FILTER_BPH -- filter bits per distinct hash -- once): bench database (configs[2] scale), eight rotating
1e6-hash samples of three kinds (bench: 200 genomes present; noise: no database hash; real: 83 k hashes, 29 % of
the references overlap).  Prints lookup kernel us (HIP events, every launch) and whole-step us.

    python scripts/probes/filter_sweep.py            # parent: runs the settings one after the other
    python scripts/probes/filter_sweep.py child      # one measurement with the current environment
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch

    from yacht_amd import synth
    from yacht_amd.engine import RefDB

    values, offsets, sample = synth.config3_device(seed=1002, n_refs=85_205, n_sample=1_000_000, device="cuda:0")
    n = offsets.numel() - 1
    mh = synth.max_hash_for_scaled(1000)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    ROT = 8
    kinds = {"bench": [sample], "noise": [], "real": []}
    for i in range(ROT):
        if i:
            kinds["bench"].append(synth.sample_device(values, offsets, seed=2000 + i, n_sample=1_000_000, n_present=200))
        noise = torch.unique(torch.randint(0, mh, (1_050_000,), generator=g, device="cuda:0", dtype=torch.int64))
        kinds["noise"].append(noise[~torch.isin(noise, values)][:1_000_000].contiguous())
        kinds["real"].append(synth.sample_device(values, offsets, seed=77 + i, n_sample=83_000, shape="real"))
    db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n)
    info = db.info()
    out = torch.zeros(3, n, dtype=torch.int32, device="cuda:0")
    res = {"filter_MB": round(info.get("filter_bytes", 0) / 1e6, 1)}

    def step(s):
        db.run_device(s.data_ptr(), s.numel(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())

    for name, ss in kinds.items():
        for _ in range(3):
            for s in ss:
                step(s)
        db.synchronize()
        db.timing()
        t0 = time.perf_counter()
        for i in range(400):
            step(ss[i % ROT])
        db.synchronize()
        el = (time.perf_counter() - t0) / 400
        tm = db.timing()
        res[name] = {"step_us": round(1e6 * el, 2), "lookup_us_events": round(1e3 * float(tm["ms_overlap_kernel"]), 2),
                     "tail_us_events": round(1e3 * float(tm["ms_exclusive_kernels"]), 2)}
    print("FILTER_BPH=%s %s" % (os.environ.get("YH_FILTER_BPH", "default"), json.dumps(res)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        # arguments: filter bits per distinct hash; "nt:2" = the same with lib/libyacht_hip_nt.so (a build_variant)
        for arg in (sys.argv[1:] or ["0", "1", "2", "3", "4", "6"]):
            variant, _, bph = arg.rpartition(":")
            env = dict(os.environ, YH_FILTER_BPH=bph)
            if variant:
                env["YACHT_HIP_LIB"] = os.path.join(ROOT, "yacht_amd", "lib", f"libyacht_hip_{variant}.so")
            print("variant", variant or "base", end=" ", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
