#!/bin/bash
# rocprofv3 kernel statistics of the sketcher (bench_sketch.py) -> gpurun_out/sketch_kernel_stats.txt, and its bench line
# usage (GPU box, repo root): bash scripts/profile_sketch.sh
set -u
ROOT=$(pwd)
timeout 600 python bench_sketch.py > gpurun_out/bench_sketch.json 2> gpurun_out/bench_sketch.err
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_sk" -- python3 "$ROOT/bench_sketch.py" --steps 10 > /dev/null 2> "$ROOT/gpurun_out/prof_sk.err"
YH_DEBUG_TUNING=1 YH_SKETCH_BYTES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_sk_bytes" -- python3 "$ROOT/bench_sketch.py" --steps 4 > /dev/null 2>> "$ROOT/gpurun_out/prof_sk.err"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$ROOT/gpurun_out/pmc_sk" -- python3 "$ROOT/bench_sketch.py" --steps 6 > /dev/null 2>> "$ROOT/gpurun_out/prof_sk.err"
cd "$ROOT"
python3 - <<'PY' > gpurun_out/sketch_kernel_stats.txt
import csv, glob
for d in ("prof_sk", "prof_sk_bytes"):
    fs = sorted(glob.glob(f"gpurun_out/{d}/**/*kernel_stats.csv", recursive=True))
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        if "k_sketch" in r["Name"]:
            name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
            # (the launches over the whole resident sequence are the long ones; the host call's are 32 MiB pieces)
            print("%-20s launches %4s  256 Mbases in one launch: %9.2f us (the longest; the others are the host call's 32 MiB pieces and the 2 Mbase parity sample)" % (
                name, r["Calls"], float(r["MaxNs"]) / 1e3))
# counters of the launches over the whole sequence (the largest grids)
fs = sorted(glob.glob("gpurun_out/pmc_sk/**/*counter_collection.csv", recursive=True))
if fs:
    import collections
    rows = [r for r in csv.DictReader(open(fs[-1])) if "k_sketch_dna_roll" in r["Kernel_Name"]]
    big = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(list)
    for r in rows:
        if int(r["Grid_Size"]) == big:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    n_win = big / 256 * 8192
    print("k_sketch_dna_roll, one launch over %.0f windows (PMC means):" % n_win)
    for k, v in sorted(acc.items()):
        print("   %-22s %14.6g   per window %8.3f" % (k, sum(v) / len(v), sum(v) / len(v) * (64 if k.startswith("SQ_INSTS") else 1) / n_win))
    print("   (SQ_INSTS_* count wave instructions: x 64 lanes / windows = lane instructions per window)")
PY
cat gpurun_out/sketch_kernel_stats.txt
rm -rf gpurun_out/prof_sk gpurun_out/prof_sk_bytes gpurun_out/pmc_sk
