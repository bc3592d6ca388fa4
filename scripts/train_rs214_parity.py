#!/usr/bin/env python3
"""`yacht train`'s core at the ONE scale the reference publishes (README.md:276: GTDB r214 representatives, 85 205
genomes), checked against the oracle port of the reference's algorithm (oracle/yacht_oracle.cpp: the node-based hash index
of main.cpp:215-246 built on one thread, the scatter of :249-312 on all host threads, the selection of :371-407) -- once,
as evidence: every kept pair (i, j, count), the three index statistics and the selection ORDER must be equal.

Synthetic rs214-shaped sketches (synth.config3_device, seed 1002: 10 % of the references in clusters of 2-8 sharing 10-95 %)
generated in HBM.  The oracle needs ~45 GB of host memory and a few minutes at this size; the script refuses to start it
with less than 96 GB free.   usage (GPU box):  python scripts/train_rs214_parity.py [n_refs] > gpurun_out/train_rs214_parity.json"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import oracle  # noqa: E402  (the checker)
from yacht_amd import synth  # noqa: E402
from yacht_amd.engine import YH_DB_PAIRWISE_ONLY, RefDB, train_select  # noqa: E402


def host_free_gib() -> float:
    with open("/proc/meminfo") as f:
        for ln in f:
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 2 ** 20
    return 0.0


def main() -> int:
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_refs = int(pos[0]) if pos else 85_205
    c = 0.95 ** 31
    values, offsets, _sample = synth.config3_device(seed=1002, n_refs=n_refs, n_sample=1000, device="cuda:0")
    torch.cuda.synchronize()
    h_values = values.cpu().numpy().view(np.uint64)
    h_offsets = offsets.cpu().numpy().astype(np.uint64)
    sizes = np.diff(h_offsets).astype(np.uint32)
    t0 = time.perf_counter()
    db = RefDB.from_device(values.data_ptr(), offsets.data_ptr(), n_refs, flags=YH_DB_PAIRWISE_ONLY)
    t1 = time.perf_counter()
    pi, pj, pc = db.pairwise(c)
    t2 = time.perf_counter()
    sel = train_select(sizes, pi, pj)
    t3 = time.perf_counter()
    stats = tuple(int(x) for x in db.index_stats())
    tm = db.timing()
    db.close()
    out = {"n_refs": n_refs, "n_hashes": int(h_values.size), "c_thresh": c,
           "gpu": {"create_device_s": round(t1 - t0, 4), "pairwise_s": round(t2 - t1, 4), "select_s": round(t3 - t2, 4),
                   "build_kernels_ms": round(float(tm["ms_db_build"]), 3), "pairwise_kernels_ms": round(float(tm["ms_pairwise_kernels"]), 3),
                   "pairs_kept": int(pi.size), "selected": int(sel.size), "stats_distinct_singletons_index": stats},
           "host_free_gib_before_oracle": round(host_free_gib(), 1)}
    if host_free_gib() < 96.0 and n_refs > 20_000:
        out["oracle"] = "skipped: less than 96 GiB of host memory available"
        print(json.dumps(out), flush=True)
        return 2
    threads = oracle.hardware_threads()
    t0 = time.perf_counter()
    wi, wj, wc, wstats = oracle.train_pairs(h_values, h_offsets, c, threads=threads)
    t1 = time.perf_counter()
    wsel = oracle.train_select(sizes, wi, wj)
    t2 = time.perf_counter()

    def digest(*arrs):
        h = hashlib.sha256()
        for a in arrs:
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()[:32]

    eq = {"pairs_equal": bool(np.array_equal(pi, wi) and np.array_equal(pj, wj) and np.array_equal(pc, wc)),
          "stats_equal": stats == tuple(int(x) for x in wstats),
          "selection_order_equal": bool(np.array_equal(sel, wsel))}
    out["oracle"] = {"kind": "port (oracle/yacht_oracle.cpp: index on 1 thread, scatter on %d)" % threads, "train_pairs_s": round(t1 - t0, 1),
                     "train_select_s": round(t2 - t1, 3), "pairs_kept": int(wi.size), "selected": int(wsel.size),
                     "stats_distinct_singletons_index": [int(x) for x in wstats]}
    out["digests"] = {"gpu_pairs": digest(pi, pj, pc), "oracle_pairs": digest(wi, wj, wc), "gpu_selection": digest(sel), "oracle_selection": digest(wsel)}
    out.update(eq)
    out["all_equal"] = all(eq.values())
    out["reference_published"] = "README.md:276: ~12 minutes, 52 GB, 64 threads for the whole `yacht train` command at this N"
    print(json.dumps(out), flush=True)
    if "--write-golden" in sys.argv and n_refs == 85_205:
        # the ORACLE's answers only (never the GPU's): what tests/test_gpu_train_rs214.py holds the shipped build to
        g = {"what": "yacht train core at N = 85 205 (README.md:276's scale): the ORACLE PORT's answers (oracle/yacht_oracle.cpp) on "
                     "synth.config3_device(seed=1002, n_refs=85205, n_sample=1000), as digests: sha256[:32] over the little-endian bytes "
                     "of the (i, j, count) uint32 arrays of the kept pairs in (i, j) order, and of the selection (uint32 ids in walk order)",
             "made_by": "scripts/train_rs214_parity.py --write-golden", "seed": 1002, "n_refs": n_refs, "n_hashes": int(h_values.size),
             "c_thresh": c, "pairs_kept": int(wi.size), "selected": int(wsel.size),
             "stats_distinct_singletons_index": [int(x) for x in wstats], "oracle_pairs_digest": digest(wi, wj, wc),
             "oracle_selection_digest": digest(wsel), "input_digest": digest(values.cpu().numpy(), offsets.cpu().numpy()),
             "oracle_train_pairs_s": round(t1 - t0, 1)}
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "golden_train_rs214.json"), "w") as f:
            json.dump(g, f, indent=1)
    return 0 if out["all_equal"] else 1


if __name__ == "__main__":
    sys.exit(main())
