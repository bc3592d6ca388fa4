# usage (GPU box): bash scripts/profile_bench.sh [stats|pmc|all]   -> gpurun_out/prof_stats, gpurun_out/pmc_*/
# rocprofv3 runs of bench.py: kernel trace + stats in one run, each --pmc group in a run of its own (gpurun
# refuses --pmc together with tracing domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
MODE=${1:-stats}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py --steps 20 --warmup 3 --min-timed-steps 300 --min-timed-ms 0 --no-cpu-baseline --no-host-inclusive --no-real-shape --no-batched --no-train --no-scaling-model"
if [ "$MODE" = stats ] || [ "$MODE" = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $B > gpurun_out/prof_stats.log 2>&1
fi
if [ "$MODE" = pmc ] || [ "$MODE" = all ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- $B > gpurun_out/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- $B > gpurun_out/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_sq1 -- $B > gpurun_out/pmc_sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc_sq2 -- $B > gpurun_out/pmc_sq2.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --output-format csv -d gpurun_out/pmc_tcc -- $B > gpurun_out/pmc_tcc.log 2>&1
fi
python3 scripts/summarize_prof.py gpurun_out
# keep what is judged (per-kernel stats, per-dispatch counter rows of OUR query kernels), drop the bulk
python3 - <<'PY'
import csv, glob, os
keep = ("k_step_fused", "k_stream_lookup", "k_index_lookup", "k_reduce_replicas", "k_excl_pieces")
for f in glob.glob("gpurun_out/pmc_*/*/*_counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    out = os.path.join("gpurun_out", os.path.basename(os.path.dirname(os.path.dirname(f))) + ".csv")
    with open(out, "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size", "VGPR_Count", "LDS_Block_Size"], extrasaction="ignore")
        w.writeheader()
        for r in rows:
            if any(k in r["Kernel_Name"] for k in keep):
                r["Kernel_Name"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                w.writerow(r)
for f in glob.glob("gpurun_out/prof_stats/*/*_kernel_stats.csv"):
    os.replace(f, "gpurun_out/kernel_stats.csv")
PY
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_tcc gpurun_out/prof_stats gpurun_out/*.log
