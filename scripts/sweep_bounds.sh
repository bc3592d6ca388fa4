#!/bin/bash
# round 6: k_piece_bounds without its zero-fill reads 0.40 GB in ~165 us (2.4 TB/s) at configs[3]: 313 workgroups of 16 waves on
# 256 CUs (a second, 22 %-full round) and two sketches of 20 dependent load rounds per wave.  Sketches per workgroup (YH_PC_SK),
# lanes per workgroup and loads in flight per wave (build variants).   usage (GPU box, repo root): bash scripts/sweep_bounds.sh
cd "$GRAFT_REPO_ROOT" || exit 1
one() {
    env YH_DEBUG_TUNING=1 "$@" python bench_train.py --device-input --no-oracle --no-scaling-model --steps 9 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['device_input']['seconds']
print('%-44s total %.3f ms  build kernels %.3f  pair kernels %.3f  golden %s' % ('$LABEL', 1e3 * s['total'], s['db_build_kernels_ms'], s['pairwise_kernels_ms'], d['full_size_vs_genuine_reference']['all_equal']))
"
}
for v in "512 4" "256 4" "256 8" "512 8" "128 8" "1024 4"; do
    set -- $v
    lib=$(python -c "from yacht_amd import build; print(build.build_variant('bnd_$1_$2', {'YH_PC_BOUND_THREADS': $1, 'YH_PC_BOUND_U': $2}))")
    for sk in 32 16 8 4; do
        LABEL="threads $1 U $2 SK $sk" one YACHT_HIP_LIB=$lib YH_PC_SK=$sk
    done
done
