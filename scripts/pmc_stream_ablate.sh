#!/bin/bash
# round 6 (VERDICT r05 "next" 7): where k_stream_lookup's bytes beyond its layout come from.  FETCH_SIZE / WRITE_SIZE / TCC_REQ per
# launch of the shipped kernel and of traffic-attribution builds (YH_ABLATE_STREAM: 1 = candidates not confirmed -- no srec / sample
# read --, 2 = the probes do not read their lane's 16 delta bytes again, 3 = both).  usage (GPU box, repo root): bash scripts/pmc_stream_ablate.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r06/pmc_stream_ablate.txt
mkdir -p gpurun_out/r06
: > $OUT
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset YACHT_HIP_LIB; else export YACHT_HIP_LIB=$(python3 -c "from yacht_amd import build; print(build.build_variant('abl_stream_$v', {'YH_ABLATE_STREAM': $v}))"); fi
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/pmcs_${v}_$tag
    rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcs_${v}_$tag -- python3 bench.py --steps 20 --warmup 5 --no-train --no-sketch --no-scaling-model --no-cpu-baseline --no-host-inclusive --no-batched --no-real-shape --min-timed-steps 40 --percentile-steps 20 > /dev/null 2>&1
  done
  rm -rf /tmp/pmcs_${v}_trace
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmcs_${v}_trace -- python3 bench.py --steps 20 --warmup 5 --no-train --no-sketch --no-scaling-model --no-cpu-baseline --no-host-inclusive --no-batched --no-real-shape --min-timed-steps 40 --percentile-steps 20 > /dev/null 2>&1
  python3 - $v >> $OUT <<'PY'
import csv, glob, collections, sys
v = sys.argv[1]
print("== YH_ABLATE_STREAM =", v)
for d in sorted(glob.glob(f"/tmp/pmcs_{v}_[A-Z]*")):
    fs = sorted(glob.glob(d + "/*/*_counter_collection.csv"))
    if not fs:
        print(d, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if "k_stream_lookup" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, x in sorted(acc.items()):
        print(f"{c:24s} {sum(x) / len(x):16.6g}   (n={len(x)})")
for f in glob.glob(f"/tmp/pmcs_{v}_trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_stream_lookup" in r["Name"]:
            print("k_stream_lookup calls", r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3))
PY
done
cat $OUT
