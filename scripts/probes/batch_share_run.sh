#!/bin/bash
# GPU box: scripts/probes/batch_share_trace.py under rocprofv3 --kernel-trace for G in "$@" (default 8 1), the steady-state
# per-block table of each (scripts/probes/batch_share_blocks.py) into gpurun_out/batch_share_G<g>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
B=${BS_B:-64}       # samples per block (round 6: up to 256)
NBLK=${BS_BLOCKS:-40}
SUF=$([ "$B" = 64 ] && echo "" || echo "_B$B")
for g in ${@:-8 1}; do
  rm -rf /tmp/bs_$g
  rocprofv3 --kernel-trace --output-format csv -d /tmp/bs_$g -- python3 scripts/probes/batch_share_trace.py $g $NBLK $B 2> gpurun_out/batch_share_G$g$SUF.err
  grep "per block" gpurun_out/batch_share_G$g$SUF.err > gpurun_out/batch_share_G$g$SUF.txt
  python3 scripts/probes/batch_share_blocks.py /tmp/bs_$g 3 >> gpurun_out/batch_share_G$g$SUF.txt
done
