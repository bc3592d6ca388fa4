# One measurement session on the GPU box (rounds 5-6; afterwards, here: bash scripts/adopt_session.sh <tag>): rocprofv3 kernel stats + PMC passes of bench.py, the bench lines, the train
# profile (host input and device input), the commands end to end.  usage: bash scripts/measure_session.sh [tag, default r06]  -> gpurun_out/
set -u
TAG=${1:-r06}
cd "$GRAFT_REPO_ROOT"
timeout 1500 bash scripts/profile_bench.sh all > gpurun_out/summary.txt 2>&1
# (the traffic files first, into profiles/ too: bench.py attaches roofline.traffic from the one whose source tag matches this
# build; make_traffic_json.py takes the database size from a bench line -- a short one of the profiled configuration)
python bench.py --steps 20 --warmup 3 --min-timed-steps 300 --min-timed-ms 0 --no-cpu-baseline --no-host-inclusive --no-real-shape --no-batched --no-train --no-scaling-model > gpurun_out/bench_for_traffic.json 2> /dev/null
python scripts/make_traffic_json.py gpurun_out gpurun_out $TAG > gpurun_out/traffic.txt 2>&1
python scripts/make_traffic_json.py gpurun_out profiles $TAG >> gpurun_out/traffic.txt 2>&1
# (the train kernels' counters next: bench_train.py -- also as bench.py's child -- attaches `traffic` from profiles/traffic_<tag>_train.json)
timeout 600 bash scripts/pmc_train.sh $TAG > gpurun_out/pmc_train.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
timeout 600 python bench_train.py > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err
timeout 300 bash scripts/profile_train.sh > gpurun_out/train_kernel_stats.txt 2>&1
TRAIN_PROF_ARGS="--device-input" timeout 300 bash scripts/profile_train.sh > gpurun_out/train_kernel_stats_device_input.txt 2>&1; rm -rf gpurun_out/prof_train
timeout 300 python scripts/probes/train_share_probe.py > gpurun_out/train_share_probe.txt 2>/dev/null
timeout 1200 python bench_e2e.py --out gpurun_out > gpurun_out/e2e.log 2> gpurun_out/e2e.err
tail -3 gpurun_out/bench_n1.err; tail -2 gpurun_out/bench_train.err; head -c 600 gpurun_out/bench_n1.json; echo; cat gpurun_out/train_share_probe.txt; head -20 gpurun_out/summary.txt; tail -5 gpurun_out/e2e.err; cat gpurun_out/e2e.log | head -c 3000
