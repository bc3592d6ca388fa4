cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_batch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive --no-real-shape --no-train --no-scaling-model > gpurun_out/prof_batch.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_batch/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    if "k_batch" in n or "k_excl_collect" in n or "fillBuffer" in n or "copyBuffer" in n:
        print(f"{n:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:9.3f}")
PY
rm -rf gpurun_out/prof_batch
