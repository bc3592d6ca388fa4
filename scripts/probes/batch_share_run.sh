#!/bin/bash
# GPU box: scripts/probes/batch_share_trace.py under rocprofv3 --kernel-trace for G in "$@" (default 8 1), the steady-state
# per-block table of each (scripts/probes/batch_share_blocks.py) into gpurun_out/batch_share_G<g>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for g in ${@:-8 1}; do
  rm -rf /tmp/bs_$g
  rocprofv3 --kernel-trace --output-format csv -d /tmp/bs_$g -- python3 scripts/probes/batch_share_trace.py $g 40 2> gpurun_out/batch_share_G$g.err
  grep "per block" gpurun_out/batch_share_G$g.err > gpurun_out/batch_share_G$g.txt
  python3 scripts/probes/batch_share_blocks.py /tmp/bs_$g 3 >> gpurun_out/batch_share_G$g.txt
done
