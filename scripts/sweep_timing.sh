# usage (GPU box): bash scripts/sweep_timing.sh  -- step time against the event-sampling period
for e in 1 4 16 0; do
  YH_TIMING_EVERY=$e python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-indexed 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('every=$e', 'step_ms', d['ms_per_step'], 'k1_ms', d['roofline']['kernel_ms_avg'], 'excl_ms', d['roofline']['exclusive_kernels_ms_avg'])"
done
