# usage: bash scripts/present_sweep.sh [variant] [wgs] [present list] [seed list]
V=${1:-base}; W=${2:-1024}; PL=${3:-"0 200 2000"}; SL=${4:-1002}
if [ "$V" = base ]; then L=$PWD/yacht_amd/lib/libyacht_hip.so; else L=$PWD/yacht_amd/lib/libyacht_hip_$V.so; fi
for sd in $SL; do for pr in $PL; do
 YACHT_HIP_LIB=$L YH_TILE_WGS=$W python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-only --present $pr --seed $sd 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V seed=$sd present=$pr', 'k1_ms', d['roofline']['kernel_ms_avg'], 'GB/s', d['roofline']['achieved'], 'sample', d['config']['sample_hashes'])"
done; done
