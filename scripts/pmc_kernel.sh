# usage (GPU box): bash scripts/pmc_kernel.sh <kernel substring[,substring...]> <out tag> -- <program and args>
# SQ and L2 counters of the kernels (means per launch), each group in a pass of its own.
K="$1"; TAG="$2"; shift 3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmck_${TAG}_$i -- "$@" > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmck_${TAG}_trace -- "$@" > /dev/null 2>&1
python3 - "$K" "$TAG" <<'PY'
import csv, glob, collections, sys
KS, TAG = sys.argv[1].split(","), sys.argv[2]
for K in KS:
    print("==", K)
    for d in sorted(glob.glob(f"gpurun_out/pmck_{TAG}_[0-9]*")):
        fs = sorted(glob.glob(d + "/*/*_counter_collection.csv"))
        if not fs: print(d, "no output"); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[-1])):
            if K in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(acc.items()):
            print(f"{c:40s} {sum(v) / len(v):14.6g}   (n={len(v)})")
for f in glob.glob(f"gpurun_out/pmck_{TAG}_trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "yh" in r["Name"] or "k_" in r["Name"]:
            print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
