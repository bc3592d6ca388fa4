#!/bin/bash
# the row pass at 85 205 (and 20 000, 40 000) references by lanes per row, count width and columns per block
# (scripts/probes/rs214_rows_probe.py); the first line of each size is the library's own choice
cd "$GRAFT_REPO_ROOT" || exit 1
export YH_DEBUG_TUNING=1
run() { n=$1; shift; echo -n "N=$n  "; env "$@" python3 scripts/probes/rs214_rows_probe.py $n 2>/dev/null | tail -1; }
run 85205 YH_PAIR_X=0
run 85205 YH_PAIR_THREADS=512
run 85205 YH_PAIR_THREADS=256 YH_PAIR_COLS=73728
run 85205 YH_PAIR_THREADS=512 YH_PAIR_COLS=73728
run 85205 YH_PAIR_THREADS=1024 YH_PAIR_COLS=73728
run 85205 YH_PAIR_THREADS=512 YH_PAIR_NO_HALF=1 YH_PAIR_COLS=36864
run 85205 YH_PAIR_NO_HALF=1
run 85205 YH_PAIR_THREADS=1024 YH_PAIR_COLS=42624
run 85205 YH_PAIR_THREADS=1024 YH_PAIR_COLS=21312
run 85205 YH_PAIR_THREADS=256 YH_PAIR_COLS=10688
run 20000 YH_PAIR_X=0
run 20000 YH_PAIR_THREADS=256
run 20000 YH_PAIR_THREADS=1024
run 40000 YH_PAIR_X=0
run 40000 YH_PAIR_THREADS=512
run 40000 YH_PAIR_THREADS=1024 YH_PAIR_COLS=20032
