# usage (GPU box): bash scripts/profile_bench.sh [stats|pmc|all]   -> gpurun_out/prof_*/
# rocprofv3 runs: kernel trace + stats in one run, each --pmc group in its own run (gpurun refuses
# mixing --pmc with tracing domains; FETCH_SIZE and WRITE_SIZE do not fit one pass).
MODE=${1:-stats}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export YH_TILE_WGS=${YH_TILE_WGS:-}
B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline"
if [ "$MODE" = stats ] || [ "$MODE" = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- $B > gpurun_out/prof_stats.log 2>&1
fi
if [ "$MODE" = pmc ] || [ "$MODE" = all ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- $B > gpurun_out/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- $B > gpurun_out/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq1 -- $B > gpurun_out/pmc_sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -- $B > gpurun_out/pmc_sq2.log 2>&1
fi
python3 scripts/summarize_prof.py gpurun_out
