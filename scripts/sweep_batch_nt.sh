#!/bin/bash
# k_batch_lookup with non-temporal loads for the read-once lines (YH_BATCH_NT build variants: 1 buckets, 2 sample hashes, 3 both):
# do they leave the XCD's L2 to the presence-filter lines the 64 samples of a quantile step share?
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
from yacht_amd import build
for v in (1, 2, 3):
    build.build_variant(f"nt{v}", {"YH_BATCH_NT": v})
PY
for v in 0 1 2 3; do
  lib="yacht_amd/lib/libyacht_hip_nt$v.so"; [ $v = 0 ] && lib="yacht_amd/lib/libyacht_hip.so"
  for g in 1 8; do echo -n "NT $v  "; YACHT_HIP_LIB=$lib python3 scripts/probes/batch_share_trace.py $g 60 2>&1 | grep "per block"; done
done
