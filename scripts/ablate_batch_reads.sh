#!/bin/bash
# round 6: what k_batch_lookup's time is made of -- timing-only builds (results wrong): 1 = the filter word is read but no bucket,
# 2 = no filter word, a bucket for every fourth hash (what a free, perfect filter would leave), 3 = neither read.
# usage (GPU box, repo root): bash scripts/ablate_batch_reads.sh
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset YACHT_HIP_LIB; else export YACHT_HIP_LIB=$(python3 -c "from yacht_amd import build; print(build.build_variant('abl_batch_$v', {'YH_ABLATE_BATCH_READS': $v}))"); fi
  echo -n "YH_ABLATE_BATCH_READS=$v  "; python3 scripts/probes/batch_share_trace.py 1 12 256 2>&1 | grep "per block"
done
# ... and what one global atomic per hit costs (no LDS table: results right) -- what a window-major order of the lookups would need
export YACHT_HIP_LIB=$(python3 -c "from yacht_amd import build; print(build.build_variant('batch_direct', {'YH_BATCH_DIRECT_ATOMICS': 1}))")
echo -n "YH_BATCH_DIRECT_ATOMICS=1  "; python3 scripts/probes/batch_share_trace.py 1 12 256 2>&1 | grep "per block"
