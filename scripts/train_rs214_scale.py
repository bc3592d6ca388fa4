#!/usr/bin/env python3
"""`yacht train` core at the scale of the reference's one published figure (README.md:276: GTDB
r214 representatives, 85 205 genomes, "around 12 minutes", 52 GB, 64 threads, whole command).
Synthetic rs214-shaped sketches generated in HBM; times yh_db_create_device (partition + index
build), yh_pairwise (dense 85 205^2 int32 block = 29 GB in HBM) and yh_train_select."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import YH_DB_PAIRWISE_ONLY, RefDB, train_select  # noqa: E402

n_refs = int(sys.argv[1]) if len(sys.argv) > 1 else 85_205
values, offsets, _sample = synth.config3_device(seed=1002, n_refs=n_refs, n_sample=1000, device="cuda:0")
sizes = (offsets[1:] - offsets[:-1]).cpu().numpy().astype(np.uint32)
torch.cuda.synchronize()
c = 0.95 ** 31
res = []
for it in range(3):
    t0 = time.perf_counter()
    db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_refs, flags=YH_DB_PAIRWISE_ONLY)
    t1 = time.perf_counter()
    pi, pj, pc = db.pairwise(c)
    t2 = time.perf_counter()
    sel = train_select(sizes, pi, pj)
    t3 = time.perf_counter()
    tm = db.timing()
    st = db.index_stats()
    info = db.info()
    db.close()
    res.append((t1 - t0, t2 - t1, t3 - t2, tm["ms_pairwise_kernels"], tm["ms_db_build"]))
b, p, s, kp, kb = (float(np.median([r[k] for r in res[1:]])) for k in range(5))
print(json.dumps({"n_refs": n_refs, "n_hashes": int(values.numel()), "distinct": st[0], "shared": st[2],
                  "postings": info["n_shared_postings"], "pairs_kept": int(pi.size), "selected": int(sel.size),
                  "seconds": {"build": round(b, 4), "pairwise": round(p, 4), "select": round(s, 4),
                              "total": round(b + p + s, 4)},
                  "pairwise_kernels_ms": round(kp, 2), "build_kernels_ms": round(kb, 2),
                  "pair_queries_per_s": round(n_refs * (n_refs - 1) / 2 / (b + p + s), 1),
                  "reference_published": "README.md:276: ~12 min, 52 GB, 64 threads (whole `yacht train` incl. ingest)"}))
