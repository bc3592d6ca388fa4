#!/bin/bash
# round 6: k_batch_lookup runs at the isolated-read ceiling (4.6e10 L2 misses/s) and its misses per sample hash ROSE with the block
# (0.51 at 64 samples per block, 0.63 at 256: profiles/r06/pmc_batch_lookup_G1_B{64,256}.txt) although more samples share every
# presence-filter line -- 16 384 workgroups each looping over eight slots drift apart, and with them the filter windows in flight.
# Workgroups of the launch: 2 048 (the resident set) ... one per slot (0).   usage (GPU box, repo root): bash scripts/sweep_batch_grid.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export YH_DEBUG_TUNING=1
for b in 256 64; do
  for g in 16384 2048 4096 8192 32768 0; do
    echo -n "B=$b grid=$g  "; YH_BATCH_GRID=$g python3 scripts/probes/batch_share_trace.py 1 12 $b 2>&1 | grep "per block"
  done
done
for g in 16384 0; do echo -n "G=8 B=256 grid=$g  "; YH_BATCH_GRID=$g python3 scripts/probes/batch_share_trace.py 8 20 256 2>&1 | grep "per block"; done
