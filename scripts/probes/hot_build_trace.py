#!/usr/bin/env python3
"""Where the build of a database with hot k-mers spends its time (YH_TRACE_BUILD=1): rs214 scale + 50 hashes in 5 000-40 000
references each, the full handle.  usage (GPU box): YH_DEBUG_TUNING=1 YH_TRACE_BUILD=1 python scripts/probes/hot_build_trace.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from hot_kmers_bench import inject  # noqa: E402
from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import YH_DB_DEFAULT, RefDB  # noqa: E402

n = 85_205
plan = synth.global_db_plan(1002, n, cluster_frac=0.10, median=3300.0, sigma=0.6, lo=300, hi=15000)
vt, ot = synth.global_db_refs_device(plan, np.arange(n), device="cuda:0")
hv, ho, hot, holders = inject(vt, ot, 50, 5000, 40000, 12)
for name, (v, o) in (("uniform", (vt, ot)), ("hot", (hv, ho)), ("hot again", (hv, ho))):
    torch.cuda.synchronize()
    print("=====", name, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    db = RefDB.from_device(v.data_ptr(), o.data_ptr(), n, flags=YH_DB_DEFAULT)
    db.synchronize()
    print("===== %s: create %.1f ms, build kernels %.1f ms" % (name, 1e3 * (time.perf_counter() - t0), db.timing()["ms_db_build"]), file=sys.stderr, flush=True)
    db.close()
