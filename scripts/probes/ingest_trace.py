#!/usr/bin/env python3
"""Where `yacht train`'s one-pass ingest spends its threads at rs214 scale: writes the synthetic 85 205-member archive of bench_e2e.py,
then reads it in FRESH processes (the command's situation: every page of the parse is touched for the first time) with the
library's phase trace on -- wall against thread-seconds of read + inflate / gunzip / parse + md5.
usage (GPU box): python scripts/probes/ingest_trace.py [refs] [threads ...]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from yacht_amd import utils, train_core
t0 = time.perf_counter()
info = utils.ingest_zip_database(sys.argv[2], sys.argv[3], 31, int(sys.argv[4]), write_files=False)
print("ingest %.3f s  %d signatures  threads %s" % (time.perf_counter() - t0, len(info), sys.argv[4]), flush=True)
train_core.drop_parsed_sketches()
"""


def main():
    import numpy as np
    import torch

    import bench_e2e
    from yacht_amd import synth

    refs = int(sys.argv[1]) if len(sys.argv) > 1 else 85_205
    threads = [int(x) for x in sys.argv[2:]] or [64, 32, 128]
    work = "/tmp/yacht_ingest_trace"
    os.makedirs(work, exist_ok=True)
    values, offsets, _ = synth.config3_device(seed=1002, n_refs=refs, n_sample=1000, device="cuda:0")
    v = values.cpu().numpy().view(np.uint64)
    o = offsets.cpu().numpy().astype(np.uint64)
    del values, offsets
    torch.cuda.empty_cache()
    z = os.path.join(work, "refs.zip")
    print("archive written in %.1f s, %.2f GB" % (bench_e2e.write_db_zip(z, v, o, 64), os.path.getsize(z) / 1e9), flush=True)
    env = dict(os.environ, YH_DEBUG_TUNING="1", YH_TRACE_BUILD="1")
    for t in threads:
        for rep in range(2):
            p = subprocess.run([sys.executable, "-c", CHILD, ROOT, z, os.path.join(work, f"w{t}_{rep}"), str(t)], env=env, capture_output=True, text=True)
            print(p.stdout.strip())
            print("\n".join(ln for ln in p.stderr.splitlines() if "[yh ingest]" in ln), flush=True)


if __name__ == "__main__":
    main()
