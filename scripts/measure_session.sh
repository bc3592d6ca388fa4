set -u
cd "$GRAFT_REPO_ROOT"
timeout 1500 bash scripts/profile_bench.sh all > gpurun_out/summary.txt 2>&1
timeout 900 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
timeout 600 python bench_train.py > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err
timeout 300 bash scripts/profile_train.sh > gpurun_out/train_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_train
timeout 300 python scripts/probes/train_share_probe.py > gpurun_out/train_share_probe.txt 2>/dev/null
python scripts/make_traffic_json.py gpurun_out gpurun_out r03 > gpurun_out/traffic.log 2>&1
tail -3 gpurun_out/bench_n1.err; tail -2 gpurun_out/bench_train.err; head -c 600 gpurun_out/bench_n1.json; echo; cat gpurun_out/train_share_probe.txt; head -20 gpurun_out/summary.txt
