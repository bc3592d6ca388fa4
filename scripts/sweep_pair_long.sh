#!/bin/bash
# k_pair_rows by the length from which a list record is walked by the whole wave (one holder per lane, one round trip) instead of
# by its own lane (one holder after the other): YH_PAIR_LONG build variants, configs[3] from HBM
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
from yacht_amd import build
for v in (0, 4, 8):
    build.build_variant(f"pl{v}", {"YH_PAIR_LONG": v})
PY
bash scripts/sweep_train_variants.sh "pl0 pl4 pl8 default" 2>&1 | tee gpurun_out/sweep_pair_long.txt
