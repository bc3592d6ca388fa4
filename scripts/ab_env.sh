#!/bin/bash
# A/B of one tuning variable on one box: `yacht train` (configs[3], sketches in HBM) with and without it, alternating.
# usage (GPU box, repo root): bash scripts/ab_env.sh YH_FZ_SORTED=1 [repeats]
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1; R=${2:-3}
one() {
    env "$@" python bench_train.py --device-input --no-oracle --no-scaling-model --steps 7 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['device_input']['seconds']
print('%-28s total %.3f ms  build kernels %.3f  pair kernels %.3f  golden %s' % ('$LABEL', 1e3 * s['total'], s['db_build_kernels_ms'], s['pairwise_kernels_ms'], d['full_size_vs_genuine_reference']['all_equal']))
"
}
for i in $(seq $R); do
    LABEL="default" one YH_DEBUG_TUNING=0
    LABEL="$V" one YH_DEBUG_TUNING=1 "$V"
done
