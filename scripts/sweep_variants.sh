# usage (GPU box): bash scripts/sweep_variants.sh "<variant names>" "<YH_TILE_WGS values>" [parity]
# variant "base" = lib/libyacht_hip.so, others = lib/libyacht_hip_<name>.so (build.py build_variant)
VARS=${1:-base}
WGS=${2:-512}
if [ "${3:-parity}" = parity ]; then python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2; fi
for v in $VARS; do
  if [ "$v" = base ]; then L=$PWD/yacht_amd/lib/libyacht_hip.so; else L=$PWD/yacht_amd/lib/libyacht_hip_$v.so; fi
  for w in $WGS; do
    YACHT_HIP_LIB=$L YH_TILE_WGS=$w python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-only 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v wgs=$w', 'step_ms', d['ms_per_step'], 'k1_ms', d['roofline']['kernel_ms_avg'], 'GB/s', d['roofline']['achieved'], 'P', d['config']['partitions'])"
  done
done
