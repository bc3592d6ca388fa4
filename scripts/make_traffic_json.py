#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernel (default k_stream_lookup) from the rocprofv3 --pmc passes.

    python scripts/make_traffic_json.py gpurun_out profiles/traffic_r01.json <n_hashes> [kernel]

MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB per dispatch; on gfx950
FETCH_SIZE reports exactly HALF of the bytes of a wide coalesced streaming read (16 B per lane),
which is what this kernel's stream is, so it is doubled; WRITE_SIZE is exact.  The two counters
do not fit one pass (TCC slots), hence two runs of the same command.
"""
import collections
import csv
import glob
import json
import os
import sys

root, out, n_hashes = sys.argv[1], sys.argv[2], int(sys.argv[3])
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "k_stream_lookup"


def mean_counter(d, counter, kernel_tag):
    files = glob.glob(os.path.join(root, d, "*", "*_counter_collection.csv"))
    files.sort(key=os.path.getmtime)
    vals = []
    for r in csv.DictReader(open(files[-1])):
        if r["Counter_Name"] == counter and kernel_tag in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


fetch_kib, nf = mean_counter("pmc_fetch", "FETCH_SIZE", KERNEL)
write_kib, nw = mean_counter("pmc_write", "WRITE_SIZE", KERNEL)
hbm = 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0
json.dump({
    "kernel": KERNEL,
    "n_hashes": n_hashes,
    "FETCH_SIZE_KiB_mean": fetch_kib, "launches_fetch": nf,
    "WRITE_SIZE_KiB_mean": write_kib, "launches_write": nw,
    "correction": "read bytes = 2 x FETCH_SIZE (gfx950, 16-B/lane coalesced stream); write bytes = WRITE_SIZE",
    "hbm_bytes_per_launch": int(hbm),
}, open(out, "w"), indent=1)
print(open(out).read())
