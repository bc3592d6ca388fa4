#!/bin/bash
# timing-only builds of k_bucket_sort<true> (YH_ABLATE_FZ; results wrong): device time of the build kernels at configs[3]
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
    lib=yacht_amd/lib/libyacht_hip.so
    [ "$v" != default ] && lib=yacht_amd/lib/libyacht_hip_$v.so
    YACHT_HIP_LIB=$PWD/$lib python bench_train.py --device-input --no-oracle --no-scaling-model --steps 7 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['device_input']['seconds']
print('%-10s build kernels %.3f ms  pair kernels %.3f' % ('$v', s['db_build_kernels_ms'], s['pairwise_kernels_ms']))
"
done
