#!/usr/bin/env python3
"""HBM traffic per launch of the `yacht train` kernels from the rocprofv3 --pmc passes of scripts/pmc_train.sh.

    python scripts/make_train_traffic_json.py gpurun_out/pmc_train_kernels.txt out.json r05

MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reports HALF of the
bytes of a wide (16 B per lane) coalesced streaming read, WRITE_SIZE is exact for streaming stores, "other access widths are
uncalibrated: calibrate on a known byte count in your own access pattern".  The calibration point here is k_piece_bounds: it
reads every hash of the database exactly once with 8-byte-per-lane coalesced loads (8 H bytes) and writes 8 H bytes of zeros
(the cleared records) + the small bounds matrix: read_factor = 8 H / (FETCH_SIZE x 1024).  The same factor is applied to the
other kernels' coalesced reads (the pairs of a bucket, the records of a sketch); `read_bytes_x1` keeps the raw figure beside it.
bench_train.py attaches `traffic` to its device-input roofline when `source_tag` matches the kernels' source."""
import datetime
import hashlib
import json
import os
import re
import subprocess
import sys

src, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = 50_000_000  # configs[3]
P_OUT = 20_000


def source_tag() -> str:
    h = hashlib.sha256()
    for f in ("yh_sort.hip", "yh_pairwise.hip", "yh_common.h"):
        with open(os.path.join(ROOT, "yacht_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


kern = {}
cur = None
for ln in open(src):
    m = re.match(r"== (\S+)", ln)
    if m:
        cur = kern.setdefault(m.group(1), {})
        continue
    m = re.match(r"(\w+)\s+([0-9.e+-]+)\s+\(n=(\d+)\)", ln)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
durs = {}
for ln in open(src):
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?(k_\w+).*?\s(\d+)\s+([0-9.]+)\s+([0-9.]+)\s*$", ln)
    if m:
        durs.setdefault(m.group(1), float(m.group(3)) / 1e3)  # us
cal = kern.get("k_piece_bounds", {})
read_factor = (8.0 * H) / (cal["FETCH_SIZE"] * 1024.0) if cal.get("FETCH_SIZE") else None
try:
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = ""
per = {}
tot_r = tot_w = 0.0
for k, c in kern.items():
    if "FETCH_SIZE" not in c:
        continue
    r1 = c["FETCH_SIZE"] * 1024.0
    w = c.get("WRITE_SIZE", 0.0) * 1024.0
    r = r1 * (read_factor or 1.0)
    per[k] = {"read_bytes": round(r), "read_bytes_x1": round(r1), "write_bytes": round(w), "hbm_bytes": round(r + w),
              "l2_requests": c.get("TCC_REQ_sum"), "l2_misses": c.get("TCC_MISS_sum"), "l2_atomics": c.get("TCC_ATOMIC_sum"),
              "lds_instructions": c.get("SQ_INSTS_LDS"), "valu_instructions": c.get("SQ_INSTS_VALU"),
              "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "wave_cycles": c.get("SQ_WAVE_CYCLES"),
              "wait_any_cycles": c.get("SQ_WAIT_ANY"), "kernel_us": durs.get(k)}
    tot_r += r
    tot_w += w
alg = 8 * H + 12 * P_OUT
doc = {"kernel_set": "yacht train, sketches in HBM (configs[3]: 10 000 sketches, 5e7 hashes): k_piece_bounds, k_piece_part, k_bucket_group5, k_pair_rows",
       "source_tag": source_tag(), "taken": datetime.date.today().isoformat(), "commit": commit, "round": tag, "n_hashes": H,
       "read_factor": round(read_factor, 4) if read_factor else None,
       "read_factor_how": "k_piece_bounds reads each of the 5e7 hashes once with 8-byte-per-lane coalesced loads: 8 H bytes / (FETCH_SIZE x 1024)",
       "per_kernel": per, "hbm_bytes_per_call": round(tot_r + tot_w), "algorithmic_bytes": alg,
       "moved_over_algorithmic": round((tot_r + tot_w) / alg, 2)}
json.dump(doc, open(out, "w"), indent=1)
