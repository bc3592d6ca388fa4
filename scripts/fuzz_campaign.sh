#!/bin/bash
# A longer randomized parity campaign (GPU box): tests/tools/fuzz_parity.py under several tuning configurations and
# tests/tools/mix_calls.py, each against the CPU oracle.  usage: bash scripts/fuzz_campaign.sh [seconds per leg, default 120]
# (the train handle of every small round takes the fused path of yh_sort.hip; two legs force its list-only records and the posting arrays)
S=${1:-120}
cd "$GRAFT_REPO_ROOT" || exit 1
run() { echo "== $*"; env "$@" timeout $((S + 200)) python tests/tools/fuzz_parity.py --seconds "$S" --seed "$SEED" 2>&1 | tail -1; }
SEED=9101 run YH_DEBUG_TUNING=0
SEED=9107 run YH_DEBUG_TUNING=1 YH_CHECK_SORT=1
SEED=9102 run YH_DEBUG_TUNING=1 YH_UPLOAD_CHUNK_MIN=1 YH_CHECK_SORT=1 YH_UPLOAD_SHARES=0.3,0.3,0.2,0.1,0.1
SEED=9103 run YH_DEBUG_TUNING=1 YH_PAIR_COLS=64 YH_PAIR_THREADS=256 YH_NO_PSORT=1 YH_PAIR_NO_HALF=1
SEED=9104 run YH_DEBUG_TUNING=1 YH_INDEX_TILE=2 YH_FILTER_MIN=1 YH_FILTER_BPH=2 YH_PAIR_THREADS=1024
SEED=9105 run YH_DEBUG_TUNING=1 YH_NO_POOL=1 YH_UPLOAD_CHUNK_MIN=1000
SEED=9108 run YH_DEBUG_TUNING=1 YH_FZ_NO_INLINE=1 YH_UPLOAD_CHUNK_MIN=1
SEED=9109 run YH_DEBUG_TUNING=1 YH_NO_FUSED_TRAIN=1 YH_PAIR_COLS=128
# round 5: the distribution without a first level is the default above; its alternatives and corners
SEED=9110 run YH_DEBUG_TUNING=1 YH_NO_PIECES=1 YH_NO_PIECES_SORT=1 YH_CHECK_SORT=1
SEED=9111 run YH_DEBUG_TUNING=1 YH_GROUP_CHAINS=1 YH_PC_P2F=0.6 YH_PC_TILE_ELEMS=700 YH_CHECK_SORT=1
SEED=9112 run YH_DEBUG_TUNING=1 YH_NO_SPILL=1 YH_PC_P2F=5 YH_CHECK_SORT=1
SEED=9113 run YH_DEBUG_TUNING=1 YH_PC_PAD=640 YH_FZ_NO_INLINE=1 YH_CHECK_SORT=1 YH_UPLOAD_CHUNK_MIN=1
SEED=9114 run YH_DEBUG_TUNING=1 YH_NO_PIN=1 YH_UPLOAD_CHUNK_MIN=100000
# round 6: the sparse row pass forced onto every train handle (default only above 28 672 references), alone and with list-only records
SEED=9117 run YH_DEBUG_TUNING=1 YH_PAIR_SPARSE=1
SEED=9118 run YH_DEBUG_TUNING=1 YH_PAIR_SPARSE=1 YH_FZ_NO_INLINE=1 YH_PAIR_SPARSE_GRID=3 YH_UPLOAD_CHUNK_MIN=1
SEED=9119 run YH_DEBUG_TUNING=1 YH_BATCH_DENSE_FINAL=1 YH_PC_SK=3
echo "== --packed YH_UPLOAD_CHUNK_MIN=1"; YH_DEBUG_TUNING=1 YH_UPLOAD_CHUNK_MIN=1 YH_CHECK_SORT=1 timeout $((S + 200)) python tests/tools/fuzz_parity.py --seconds "$S" --seed 9115 --packed 2>&1 | tail -1
echo "== --packed (host-side unpacking of small databases)"; timeout $((S + 200)) python tests/tools/fuzz_parity.py --seconds $((S / 3 + 5)) --seed 9116 --packed 2>&1 | tail -1
echo "== mix_calls 150 rounds"; timeout 900 python tests/tools/mix_calls.py 150 9106 2>&1 | tail -1
