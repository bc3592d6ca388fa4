#!/bin/bash
# `yacht train` (configs[3], sketches in HBM) under tuning builds of the bucket size / workgroup width of yh_sort.hip
# (yacht_amd.build.build_variant: lib/libyacht_hip_<name>.so).  usage (GPU box, repo root): bash scripts/sweep_bucket.sh name ...
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
    lib=yacht_amd/lib/libyacht_hip.so
    [ "$v" != default ] && lib=yacht_amd/lib/libyacht_hip_$v.so
    YACHT_HIP_LIB=$PWD/$lib python bench_train.py --device-input --no-oracle --no-scaling-model --steps 7 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['device_input']['seconds']
print('%-10s total %.3f ms  create %.3f  pairwise %.3f  build kernels %.3f  pair kernels %.3f  golden %s' % ('$v', 1e3 * s['total'], 1e3 * s['create_device'], 1e3 * s['pairwise'], s['db_build_kernels_ms'], s['pairwise_kernels_ms'], d['full_size_vs_genuine_reference']['all_equal']))
"
done
