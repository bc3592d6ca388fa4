#!/bin/bash
# The batched lookup by tile size (YH_BATCH_TILE = 512, 1024, 2048, 4096: build variants): rank 0's share of a G-way hash-range
# run through dist.BatchedRangeRunner (scripts/probes/batch_share_trace.py), G = 8, 4, 2, 1 -- a rank's share of a block at G = 8
# is 3 970 tiles of 2 048 hashes for 2 048 resident workgroups.
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
from yacht_amd import build
for t in (512, 1024, 4096):
    build.build_variant(f"bt{t}", {"YH_BATCH_TILE": t})
PY
for t in 512 1024 2048 4096; do
  lib="yacht_amd/lib/libyacht_hip_bt$t.so"; [ $t = 2048 ] && lib="yacht_amd/lib/libyacht_hip.so"
  for g in 8 4 2 1; do
    echo -n "tile $t  "; YACHT_HIP_LIB=$lib python3 scripts/probes/batch_share_trace.py $g 60 2>&1 | grep "per block"
  done
done
