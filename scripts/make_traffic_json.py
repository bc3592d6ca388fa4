#!/usr/bin/env python3
"""HBM traffic per launch of the two lookup kernels from the rocprofv3 --pmc passes of scripts/profile_bench.sh.

    python scripts/make_traffic_json.py gpurun_out profiles <round tag, e.g. r02>

Writes profiles/traffic_<tag>_stream.json (k_stream_lookup) and profiles/traffic_<tag>_index.json
(k_index_lookup).  bench.py attaches `hbm_bytes_per_launch` to its roofline block only when the file's
`source_tag` (sha256 of csrc/yh_query.hip + yh_common.h) and `n_hashes` match the run.

MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950 FETCH_SIZE
reports exactly HALF of the bytes of a wide coalesced streaming read (16 B per lane) -- what k_stream_lookup's
stream is -- so it is doubled there; WRITE_SIZE is exact.  k_index_lookup reads isolated 64-byte buckets: its
requests are 64-byte ones and are counted as such (checked: TCC_MISS x 64 B agrees with FETCH_SIZE x 1 within
a few per cent in profiles/<tag>/summary.txt), so no doubling.  The two counters do not fit one pass.
"""
import collections
import csv
import datetime
import glob
import hashlib
import json
import os
import subprocess
import sys

root, out_dir, tag = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_tag() -> str:
    h = hashlib.sha256()
    for f in ("yh_query.hip", "yh_common.h"):
        with open(os.path.join(ROOT, "yacht_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def mean_counter(d, counter, kernel_tag):
    files = glob.glob(os.path.join(root, d, "*", "*_counter_collection.csv"))
    files.sort(key=os.path.getmtime)
    if not files:  # the slimmed per-pass file scripts/profile_bench.sh leaves behind
        files = [os.path.join(root, d + ".csv")]
    vals = []
    for r in csv.DictReader(open(files[-1])):
        if r["Counter_Name"] == counter and kernel_tag in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)


def stream_isolated_kib():
    """FETCH_SIZE (KiB per launch) of k_stream_lookup's ISOLATED reads -- the probes' second look at their lane's 16 delta bytes and
    the candidates' confirmation reads -- from the traffic-attribution builds of scripts/pmc_stream_ablate.sh (round 6): the shipped
    kernel's FETCH_SIZE minus that of the build without either.  Those are 64-byte requests, counted as such (x 1); only the
    coalesced stream under them is reported at half its bytes.  None when the ablation file is not there."""
    for cand in (os.path.join(root, "r06", "pmc_stream_ablate.txt"), os.path.join(root, "pmc_stream_ablate.txt"),
                 os.path.join(ROOT, "profiles", "r06", "pmc_stream_ablate.txt")):
        if os.path.exists(cand):
            fetch, v = {}, None
            for ln in open(cand):
                if ln.startswith("== YH_ABLATE_STREAM ="):
                    v = int(ln.split("=")[-1])
                elif ln.startswith("FETCH_SIZE") and v is not None:
                    fetch[v] = float(ln.split()[1])
            if 0 in fetch and 3 in fetch:
                return max(fetch[0] - fetch[3], 0.0), cand
    return None


try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = ""
bench_path = os.path.join(root, "bench_for_traffic.json")
if not os.path.exists(bench_path):
    bench_path = os.path.join(root, "bench_n1.json")  # (re-deriving from the slim files kept under profiles/<tag>/)
bench = json.load(open(bench_path))
n_hashes = int(bench["config"]["ref_hashes_per_gpu"])
for kernel, name, factor, why in (
        ("k_stream_lookup", "stream", 2.0, "read bytes = 2 x FETCH_SIZE (gfx950, 16-B/lane coalesced stream); write bytes = WRITE_SIZE"),
        ("k_index_lookup", "index", 1.0, "read bytes = FETCH_SIZE (isolated 64-byte bucket reads: 64-byte requests); write bytes = WRITE_SIZE"),
        ("k_step_fused", "fused", 1.0, "read bytes = FETCH_SIZE (isolated 64-byte reads of the lookup role + the two small tail roles); write bytes = WRITE_SIZE")):
    fetch_kib, nf = mean_counter("pmc_fetch", "FETCH_SIZE", kernel)
    write_kib, nw = mean_counter("pmc_write", "WRITE_SIZE", kernel)
    if not nf:
        continue
    if name == "index" and mean_counter("pmc_fetch", "FETCH_SIZE", "k_index_lookup_tile")[1]:
        kernel = "k_index_lookup_tile"  # (the form large samples take; bench.py names the same one)
    rec = {"kernel": kernel, "n_hashes": n_hashes, "source_tag": source_tag(), "commit": commit,
           "taken": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
           "FETCH_SIZE_KiB_mean": fetch_kib, "launches_fetch": nf, "WRITE_SIZE_KiB_mean": write_kib, "launches_write": nw,
           "correction": why, "hbm_bytes_per_launch": int(factor * fetch_kib * 1024.0 + write_kib * 1024.0)}
    iso = stream_isolated_kib() if name == "stream" else None
    if iso is not None and iso[0] < fetch_kib:
        # the blanket doubling also doubled the isolated 64-byte reads that are counted in full (rounds 2-5: 444.5 MB = 1.28 x the layout)
        rec["hbm_bytes_per_launch_blanket_doubling"] = rec["hbm_bytes_per_launch"]
        rec["isolated_FETCH_SIZE_KiB"] = iso[0]
        rec["hbm_bytes_per_launch"] = int((2.0 * (fetch_kib - iso[0]) + iso[0] + write_kib) * 1024.0)
        rec["correction"] = ("read bytes = 2 x (FETCH_SIZE - isolated) + isolated: the coalesced 16-B/lane stream is reported at half its bytes "
                             "(gfx950), the isolated 64-byte reads (probe re-reads + candidate confirmations, from the ablation builds of "
                             "scripts/pmc_stream_ablate.sh: %s) in full; write bytes = WRITE_SIZE" % os.path.relpath(iso[1], ROOT))
    req, nr = mean_counter("pmc_tcc", "TCC_REQ_sum", kernel)
    miss, _ = mean_counter("pmc_tcc", "TCC_MISS_sum", kernel)
    if nr:  # L2 requests per launch: what a kernel of isolated reads is bound by
        rec["l2_requests_per_launch"] = int(req)
        rec["l2_misses_per_launch"] = int(miss)
    path = os.path.join(out_dir, f"traffic_{tag}_{name}.json")
    json.dump(rec, open(path, "w"), indent=1)
    print(path, rec["hbm_bytes_per_launch"])
