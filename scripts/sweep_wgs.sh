# usage: bash scripts/sweep_wgs.sh  (on the GPU box)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for w in 512 1024 2048 4096; do
  echo "YH_TILE_WGS=$w"
  YH_TILE_WGS=$w python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['achieved'])"
done
python bench.py --steps 100 --warmup 10 2>/dev/null | tail -1
