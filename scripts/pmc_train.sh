#!/bin/bash
# HBM traffic + SQ / L2 counters of the `yacht train` kernels that ship (configs[3], sketches in HBM), each counter group in
# a rocprofv3 pass of its own (scripts/pmc_kernel.sh), then profiles/traffic_<tag>_train.json.
# usage (GPU box, repo root): bash scripts/pmc_train.sh r05
TAG=${1:-r05}
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/pmc_kernel.sh "k_piece_bounds,k_piece_part,k_bucket_group5,k_pair_rows" train -- python3 bench_train.py --device-input --no-oracle --no-scaling-model --steps 3 > gpurun_out/pmc_train_kernels.txt 2>&1
python3 scripts/make_train_traffic_json.py gpurun_out/pmc_train_kernels.txt gpurun_out/traffic_${TAG}_train.json $TAG
cp gpurun_out/traffic_${TAG}_train.json profiles/traffic_${TAG}_train.json 2>/dev/null
rm -rf gpurun_out/pmck_train_*
cat gpurun_out/traffic_${TAG}_train.json
