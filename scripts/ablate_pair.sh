#!/bin/bash
# k_pair_rows taken apart (YH_ABLATE_PAIR build variants: 1 no record pass, 2 no row clear / survivor scan, 4 records read but
# not added, 8 records read but not decoded), configs[3] from HBM
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
from yacht_amd import build
for v in (1, 2, 3, 4, 8, 10):
    build.build_variant(f"ap{v}", {"YH_ABLATE_PAIR": v})
PY
bash scripts/sweep_train_variants.sh "default ap1 ap2 ap3 ap4 ap8 ap10" 2>&1 | tee gpurun_out/ablate_pair.txt
