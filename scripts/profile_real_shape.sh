cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_real -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-inclusive --no-batched --no-train --no-scaling-model > gpurun_out/prof_real.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_real/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "k_step_fused" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import collections
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"].split("(")[0][-28:], r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    v.sort()
    print(k, "n", len(v), "median_us %.2f" % v[len(v) // 2], "min %.2f" % v[0])
# gaps between consecutive launches of the most common small-grid kernel
PY
rm -rf gpurun_out/prof_real
