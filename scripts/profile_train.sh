#!/bin/bash
# rocprofv3 kernel statistics of `yacht train` at configs[3] (bench_train.py, 3 timed passes, no CPU legs)
# usage (on the GPU box, from the repo root): bash scripts/profile_train.sh   -> gpurun_out/prof_train/, gpurun_out/train_stats.txt
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_train" -o train -- python3 "$ROOT/bench_train.py" --no-oracle --no-scaling-model --steps 3 ${TRAIN_PROF_ARGS:-} > "$ROOT/gpurun_out/train_prof_bench.json" 2> "$ROOT/gpurun_out/train_prof.err"
cd "$ROOT"
python3 - <<'PY' > gpurun_out/train_stats.txt
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'%':>6s}")
for r in rows[:30]:
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{nm:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:9.3f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
PY
cat gpurun_out/train_stats.txt
# per-launch durations of k_pair_rows in launch order (tuning: YH_PAIR_PROBE=2 repeats the launch)
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_train/**/*kernel_trace.csv", recursive=True))
if f:
    rows = [r for r in csv.DictReader(open(f[-1])) if "k_pair_rows" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print("k_pair_rows launches (us):", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in rows))
PY
