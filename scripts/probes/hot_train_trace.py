#!/usr/bin/env python3
"""configs[3] + 50 hot hashes: the train handle created five times (for rocprofv3 --kernel-trace --stats: which kernels the
side list costs).  usage: rocprofv3 --kernel-trace --stats -d /tmp/p -- python3 scripts/probes/hot_train_trace.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

from hot_kmers_bench import inject  # noqa: E402
from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import YH_DB_PAIRWISE_ONLY, RefDB  # noqa: E402

v, o = synth.config4(seed=1003, n_clusters=2000, size=5000)
vt = torch.from_numpy(v.view(np.int64)).to("cuda:0")
ot = torch.from_numpy(o.astype(np.int64)).to("cuda:0")
n = o.size - 1
hv, ho, hot, holders = inject(vt, ot, 50, 5000, 10000, 11)
for _ in range(5):
    db = RefDB.from_device(hv.data_ptr(), ho.data_ptr(), n, flags=YH_DB_PAIRWISE_ONLY)
    db.synchronize()
    print(db.timing()["ms_db_build"], db.info()["n_spilled_pairs"], file=sys.stderr)
    db.close()
