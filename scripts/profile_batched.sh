# rocprofv3 of the batched run (yh_run_batch_device, 64 distinct 1e6-hash samples per call at rs214 scale): kernel stats, then the
# L2 counters and FETCH_SIZE of k_batch_lookup in runs of their own.  usage (GPU box): bash scripts/profile_batched.sh > gpurun_out/batched_profile.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 10 --warmup 2 --min-timed-steps 100 --min-timed-ms 0 --no-cpu-baseline --no-host-inclusive --no-real-shape --no-train --no-scaling-model"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_batch -- $B > gpurun_out/prof_batch.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --output-format csv -d gpurun_out/pmc_batch_tcc -- $B > gpurun_out/pmc_batch_tcc.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_batch_fetch -- $B > gpurun_out/pmc_batch_fetch.log 2>&1
python3 - <<'PY'
import collections, csv, glob
f = sorted(glob.glob("gpurun_out/prof_batch/**/*kernel_stats.csv", recursive=True))[-1]
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s}")
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    if "k_batch" in n:
        print(f"{n:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:9.3f}")
for d in ("pmc_batch_tcc", "pmc_batch_fetch"):
    files = sorted(glob.glob(f"gpurun_out/{d}/*/*_counter_collection.csv"))
    if not files:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[-1])):
        if "k_batch" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, vals in sorted(v.items()):
            print(f"  {k:28s} {c:18s} n={len(vals):3d} mean per launch = {sum(vals)/len(vals):.6g}")
PY
rm -rf gpurun_out/prof_batch gpurun_out/pmc_batch_tcc gpurun_out/pmc_batch_fetch gpurun_out/*.log
