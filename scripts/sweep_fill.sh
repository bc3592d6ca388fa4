#!/bin/bash
# round 6: k_bucket_group5's inserts are the probing of a 4 096-slot table at load 0.4-0.6 (a wave waits for its unluckiest lane).
# Fewer pairs per bucket = a lower load and more buckets (more table clears, more workgroups): bucket fill in eighths of the
# capacity x slots requested before the bucket's count is known.   usage (GPU box, repo root): bash scripts/sweep_fill.sh
cd "$GRAFT_REPO_ROOT" || exit 1
one() {
    env YH_DEBUG_TUNING=1 "$@" python bench_train.py --device-input --no-oracle --no-scaling-model --steps 9 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['device_input']['seconds']
print('%-44s total %.3f ms  build kernels %.3f  pair kernels %.3f  golden %s' % ('$LABEL', 1e3 * s['total'], s['db_build_kernels_ms'], s['pairwise_kernels_ms'], d['full_size_vs_genuine_reference']['all_equal']))
"
}
for v in "5 2" "4 2" "3 2" "3 1" "4 1" "6 2" "6 3" "2 1"; do
    set -- $v
    lib=$(python -c "from yacht_amd import build; print(build.build_variant('fill_$1_$2', {'YH_BKT_FILL8': $1, 'YH_GROUP_SPEC_ITEMS': $2}))")
    LABEL="fill $1/8 spec items $2" one YACHT_HIP_LIB=$lib
done
