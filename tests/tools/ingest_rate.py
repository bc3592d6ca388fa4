#!/usr/bin/env python3
"""Ingest rate of the library's threaded .sig reader (yh_sig_batch_*) next to the json-module reader.

    python tests/tools/ingest_rate.py [n_files] [mins_per_file]
Writes synthetic sourmash-style files (mins + abundances) to a temp dir, reads them back with 1..N host
threads, checks both readers agree, prints one JSON line."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yacht_amd import train_core  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_mins = int(sys.argv[2]) if len(sys.argv) > 2 else 3300
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
rng = np.random.default_rng(0)
paths = []
for i in range(n_files):
    m = np.unique(rng.integers(0, 18446744073709552, size=n_mins, dtype=np.uint64))
    p = os.path.join(d, f"s{i}.sig")
    with open(p, "w") as f:
        json.dump([{"class": "sourmash_signature", "name": f"g{i}", "signatures": [
            {"num": 0, "ksize": 31, "seed": 42, "max_hash": 18446744073709552, "mins": [int(x) for x in m],
             "abundances": [int(a) for a in rng.integers(1, 5, size=len(m))], "molecule": "dna"}], "version": 0.4}], f)
    paths.append(p)
out = {"files": n_files, "mins_per_file": n_mins, "cores": os.cpu_count(), "native_s": {}}
ref = None
for th in (1, 8, 32, 128):
    if th > (os.cpu_count() or 1):
        break
    t0 = time.perf_counter()
    v, o = train_core.read_sketches_csr(paths, threads=th)
    out["native_s"][str(th)] = round(time.perf_counter() - t0, 3)
    ref = (v, o)
t0 = time.perf_counter()
py = train_core.read_sketches(paths[:500], 1)
out["python_json_1_thread_s_per_1000_files"] = round((time.perf_counter() - t0) * 2, 3)
out["readers_agree"] = all(np.array_equal(a, ref[0][int(ref[1][i]):int(ref[1][i + 1])]) for i, a in enumerate(py))
shutil.rmtree(d)
print(json.dumps(out))
