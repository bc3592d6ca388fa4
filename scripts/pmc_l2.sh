# usage (GPU box): bash scripts/pmc_l2.sh   -- L2 hit/miss counters of k_stream_lookup
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-indexed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d gpurun_out/pmcx_l2 -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pmcx_l2b -- $B > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcx_l2*")):
    fs = sorted(glob.glob(d + "/*/*_counter_collection.csv"))
    if not fs: print(d, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if "k_stream_lookup" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print(c, "%.5g" % (sum(v) / len(v)))
PY
