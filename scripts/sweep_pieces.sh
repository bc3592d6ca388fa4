#!/bin/bash
# `yacht train` (configs[3], sketches in HBM): the distribution without a first level (yh_sort.hip: k_piece_*) against the
# two-level one (YH_NO_PIECES=1) and over its geometry knobs.  usage (GPU box, repo root): bash scripts/sweep_pieces.sh
cd "$GRAFT_REPO_ROOT" || exit 1
one() {
    env YH_DEBUG_TUNING=1 "$@" python bench_train.py --device-input --no-oracle --no-scaling-model --steps 7 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['device_input']['seconds']
print('%-44s total %.3f ms  build kernels %.3f  pair kernels %.3f  frac %.4f  golden %s' % ('$LABEL', 1e3 * s['total'], s['db_build_kernels_ms'], s['pairwise_kernels_ms'], d['device_input']['roofline']['frac'], d['full_size_vs_genuine_reference']['all_equal']))
"
}
LABEL="two levels (YH_NO_PIECES=1)" one YH_NO_PIECES=1
for f in ${PC_P2F:-1 1.5 2 3 4 7.3}; do
    LABEL="pieces P2F=$f" one YH_PC_P2F=$f
done
for t in ${PC_TILE:-2048 3584}; do
    LABEL="pieces tile_elems=$t" one YH_PC_TILE_ELEMS=$t
done
LABEL="two levels (YH_NO_PIECES=1)" one YH_NO_PIECES=1
LABEL="pieces default" one YH_X=0
