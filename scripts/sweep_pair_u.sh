#!/bin/bash
# k_pair_rows by record loads in flight per lane (YH_PAIR_U: build variants) and lanes per row (YH_PAIR_THREADS), configs[3] from HBM
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
from yacht_amd import build
for v in (2, 6, 8, 12):
    build.build_variant(f"pu{v}", {"YH_PAIR_U": v})
PY
bash scripts/sweep_train_variants.sh "pu2 default pu6 pu8 pu12 pu8:YH_PAIR_THREADS=256 pu12:YH_PAIR_THREADS=256 pu2:YH_PAIR_THREADS=1024 default:YH_PAIR_THREADS=1024 pu6:YH_PAIR_THREADS=1024" 2>&1 | tee gpurun_out/sweep_pair_u.txt
