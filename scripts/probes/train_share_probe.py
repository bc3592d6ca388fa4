#!/usr/bin/env python3
"""GPU box: where one rank's share of a G-way hash-range `yacht train` (configs[3]) spends its time: handle creation
(upload + validation + sort + index) against the pairwise pass, wall clock and kernel events, for G = 1, 2, 4, 8."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yacht_amd import dist as ydist, synth  # noqa: E402
from yacht_amd.engine import YH_DB_PAIRWISE_ONLY, RefDB  # noqa: E402

values, offsets = synth.config4()
for G in (1, 2, 4, 8):
    b = ydist.hash_range_bounds(int(values.max()), G)
    v, o = ydist.slice_csr_to_hash_range(values, offsets, b[0], b[1])
    rows = []
    for it in range(5):
        t0 = time.perf_counter()
        db = RefDB(v, o, flags=YH_DB_PAIRWISE_ONLY)
        t1 = time.perf_counter()
        pi, pj, pc = db.pairwise(0.0)
        t2 = time.perf_counter()
        tm = db.timing()
        db.close()
        t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, tm["ms_db_build"], tm["ms_pairwise_kernels"]))
    r = np.median(np.array(rows[1:]), axis=0)
    print(json.dumps({"G": G, "hashes": int(v.size), "create_ms": round(1e3 * r[0], 3), "pairwise_ms": round(1e3 * r[1], 3),
                      "close_ms": round(1e3 * r[2], 3), "build_kernels_ms": round(float(r[3]), 3),
                      "pairwise_kernels_ms": round(float(r[4]), 3), "pairs": int(pi.size)}), flush=True)
