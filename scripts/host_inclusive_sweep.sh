B="python bench.py --no-real-shape --no-batched --no-cpu-baseline --steps 300 --percentile-steps 300"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["host_inclusive"]; print(sys.argv[1], "packed", h["ms_per_step"], h["median_ms"], h["p90_ms"], "GB/s", h["h2d_GBps"], "raw", h["raw_dense"]["ms_per_step"], "value_ms", d["ms_per_step"])'
YH_DEBUG_TUNING=1 YH_ONE_UPLOAD_STREAM=1 $B --host-depth 3 2>/dev/null | python -c "$P" one-d3
$B --host-depth 3 2>/dev/null | python -c "$P" two-d3
$B --host-depth 4 2>/dev/null | python -c "$P" two-d4
$B --host-depth 2 2>/dev/null | python -c "$P" two-d2
