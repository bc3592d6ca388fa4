#!/bin/bash
# round 6: the train side after "every pair writes its record" (no zero-fill in the bounds pass) and the sparse row pass --
# parity tests, the row pass at rs214 scale sparse vs dense, kernel statistics at rs214 scale and at configs[3]
# usage (GPU box, repo root): bash scripts/r06_train_measure.sh   -> gpurun_out/r06/
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
python3 -m pytest tests/test_gpu_pairwise.py tests/test_gpu_train_rs214.py tests/test_gpu_keys.py tests/test_gpu_golden.py -x -q > "$OUT/tests_train.txt" 2>&1
echo "tests rc=$?" | tee -a "$OUT/tests_train.txt"
tail -4 "$OUT/tests_train.txt"
export YH_DEBUG_TUNING=1
{
  for n in 85205 40000 20000; do
    echo -n "N=$n sparse(default)  "; python3 scripts/probes/rs214_rows_probe.py $n 2>/dev/null | tail -1
    echo -n "N=$n dense            "; YH_PAIR_SPARSE=0 python3 scripts/probes/rs214_rows_probe.py $n 2>/dev/null | tail -1
  done
  echo -n "N=10000 sparse forced  "; YH_PAIR_SPARSE=1 python3 scripts/probes/rs214_rows_probe.py 10000 2>/dev/null | tail -1
  echo -n "N=10000 dense          "; python3 scripts/probes/rs214_rows_probe.py 10000 2>/dev/null | tail -1
  for g in 512 2048 4096; do echo -n "N=85205 sparse grid $g  "; YH_PAIR_SPARSE_GRID=$g python3 scripts/probes/rs214_rows_probe.py 85205 2>/dev/null | tail -1; done
} > "$OUT/rs214_rows_sparse.txt" 2>&1
cat "$OUT/rs214_rows_sparse.txt"
unset YH_DEBUG_TUNING
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_rs214_train" -o t -- python3 "$ROOT/scripts/probes/rs214_rows_probe.py" 85205 > "$OUT/prof_rs214_train.out" 2>&1
cd "$ROOT"
python3 - <<'PY' > "$OUT/train_rs214_kernel_stats.txt"
import csv, glob
f = sorted(glob.glob("gpurun_out/r06/prof_rs214_train/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'%':>6s}")
for r in rows[:30]:
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{nm:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:9.3f} {100*float(r['TotalDurationNs'])/tot:6.1f}")
PY
cat "$OUT/train_rs214_kernel_stats.txt"
rm -rf "$OUT/prof_rs214_train"
bash scripts/profile_train.sh > /dev/null 2>&1
cp gpurun_out/train_stats.txt "$OUT/train_kernel_stats_cfg3.txt"; cat "$OUT/train_kernel_stats_cfg3.txt" | head -14
python3 bench_train.py --steps 5 > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"; echo "bench_train rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06/bench_train.json"))
print({k: d.get(k) for k in ("value", "seconds", "parity_bit_exact")})
print("device_input", (d.get("device_input") or {}).get("seconds"), (d.get("device_input") or {}).get("roofline"))
PY
