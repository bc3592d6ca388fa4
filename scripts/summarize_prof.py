#!/usr/bin/env python3
"""Summarize rocprofv3 CSV output of scripts/profile_bench.sh (kernel stats + PMC means per kernel)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
OURS = ("k_step_fused", "k_unpack_sample", "k_compact_rows", "k_range_mask", "k_tile_lookup", "k_stream_lookup", "k_resolve_stream", "k_wg_", "k_batch", "k_index_lookup", "k_resolve_hits", "k_excl", "k_prep", "k_mask_bits", "k_reduce_replicas", "k_sample_bounds", "k_mask_from", "k_pair", "k_overlap_bsearch",
        "k_scan_u32", "k_idx", "k_split", "k_part_scan", "k_scatter", "k_scan_refs", "k_fill", "k_bounds")


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    for tag in ("<OverlapHit>", "<FlagHit>"):
        if "k_tile_lookup" in name and tag[1:-1] in name:
            return "k_tile_lookup" + tag
    return name.split("(")[0][:60]


stats = glob.glob(os.path.join(root, "prof_stats", "*", "*_kernel_stats.csv"))
if stats:
    stats.sort(key=os.path.getmtime, reverse=True)
    print("== kernel stats (ours) ==")
    print(f"{'kernel':42s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>9s}")
    for r in csv.DictReader(open(stats[0])):
        if any(t in r["Name"] for t in OURS):
            print(f"{short(r['Name']):42s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} "
                  f"{float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:9.3f}")

for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    files.sort(key=os.path.getmtime)
    for r in csv.DictReader(open(files[-1])):
        if any(t in r["Kernel_Name"] for t in ("k_step_fused", "k_tile_lookup", "k_stream_lookup", "k_index_lookup", "k_reduce_replicas", "k_excl_pieces")):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {os.path.basename(d)} (mean per launch) ==")
    for k, v in acc.items():
        print("  " + k)
        for c, vals in sorted(v.items()):
            print(f"     {c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.6g}")

# ---- timeline of the last complete step (kernel trace): start offset, duration, gap to the previous kernel
trace = glob.glob(os.path.join(root, "prof_stats", "*", "*_kernel_trace.csv"))
if trace:
    trace.sort(key=os.path.getmtime, reverse=True)
    rows = list(csv.DictReader(open(trace[0])))
    rows = [r for r in rows if any(t in r["Kernel_Name"] for t in OURS) or "Memset" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower()]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for tag, title in (("k_step_fused", "fused step (k_step_fused: lookup k + reduce k-1 + exclusive pass k-2 in one launch)"),
                       ("k_stream_lookup", "streaming step"), ("k_index_lookup", "sample-driven step, three launches (k_index_lookup_tile)")):
        idx = [i for i, r in enumerate(rows) if tag in r["Kernel_Name"]]
        # two consecutive launches of the same lookup kernel with only step kernels between them
        pairs = [(a, b) for a, b in zip(idx, idx[1:]) if b - a <= 4]
        if not pairs:
            continue
        a, b = pairs[len(pairs) // 2]
        t0 = int(rows[a]["Start_Timestamp"])
        prev_end = t0
        print(f"== timeline of one {title} (us) ==")
        for r in rows[a:b]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print(f"  +{(s - t0)/1e3:8.2f}  dur {(e - s)/1e3:8.2f}  gap {(s - prev_end)/1e3:7.2f}  {short(r['Kernel_Name'])}")
            prev_end = e
        print(f"  step span {(int(rows[b]['Start_Timestamp']) - t0)/1e3:.2f} us")
