#!/bin/bash
# the row pass of `yacht train` (configs[3], sketches in HBM) under tuning builds (yacht_amd.build.build_variant) and widths
# usage (GPU box, repo root): bash scripts/sweep_pair.sh "512 1024" default w0 ...
cd "$GRAFT_REPO_ROOT" || exit 1
widths=$1; shift
for v in "$@"; do
  for t in $widths; do
    lib=yacht_amd/lib/libyacht_hip.so
    [ "$v" != default ] && lib=yacht_amd/lib/libyacht_hip_$v.so
    YH_DEBUG_TUNING=1 YH_PAIR_THREADS=$t YACHT_HIP_LIB=$PWD/$lib python bench_train.py --device-input --no-oracle --no-scaling-model --steps 7 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['device_input']['seconds']
print('%-8s threads %4d  pair kernels %.3f ms  golden %s' % ('$v', $t, s['pairwise_kernels_ms'], d['full_size_vs_genuine_reference']['all_equal']))
"
  done
done
