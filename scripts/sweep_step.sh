# usage (GPU box): bash scripts/sweep_step.sh "<variants>" "<YH_TILE_WGS values>"  -- the default (delta stream) layout, whole step
for v in ${1:-base}; do
  if [ "$v" = base ]; then L=$PWD/yacht_amd/lib/libyacht_hip.so; else L=$PWD/yacht_amd/lib/libyacht_hip_$v.so; fi
  for w in ${2:-512}; do
    YACHT_HIP_LIB=$L YH_TILE_WGS=$w python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-indexed 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v wgs=$w', 'step_ms', d['ms_per_step'], 'k1_ms', d['roofline']['kernel_ms_avg'], 'excl_ms', d['roofline']['exclusive_kernels_ms_avg'])"
  done
done
