# usage (GPU box): bash scripts/pmc_sq.sh <tag>   -- SQ counters of one bench run into gpurun_out/pmc_<tag>_{sq1,sq2}/ (YACHT_HIP_LIB honoured)
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-indexed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmcx_${TAG}_sq1 -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmcx_${TAG}_sq2 -- $B > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmcx_${TAG}_sq*")):
    f = sorted(glob.glob(d + "/*/*_counter_collection.csv"))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_stream_lookup" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print("${TAG}", c, "%.4g" % (sum(v) / len(v)))
PY
