#!/bin/bash
# round 6: k_bucket_group5 writes 0.71 GB for 0.40 GB of records -- half-written lines pushed out of the XCD's L2 by the pairs that
# stream through it.  Non-temporal loads of the pairs (1) / non-temporal stores of the pairs in k_piece_part (2) / both (3):
# kernel times and WRITE_SIZE / FETCH_SIZE of the grouping pass.   usage (GPU box, repo root): bash scripts/sweep_group_nt.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in 0 1 2 3; do
  if [ $v = 0 ]; then unset YACHT_HIP_LIB; else export YACHT_HIP_LIB=$(python3 -c "from yacht_amd import build; print(build.build_variant('group_nt_$v', {'YH_GROUP_NT': $v}))"); fi
  echo "== YH_GROUP_NT = $v"
  python3 bench_train.py --device-input --no-oracle --no-scaling-model --steps 9 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['device_input']['seconds']
print('total %.3f ms  build kernels %.3f  pair kernels %.3f  golden %s' % (1e3 * s['total'], s['db_build_kernels_ms'], s['pairwise_kernels_ms'], d['full_size_vs_genuine_reference']['all_equal']))
"
  for grp in WRITE_SIZE FETCH_SIZE; do
    rm -rf /tmp/gnt_${v}_$grp
    rocprofv3 --pmc $grp --output-format csv -d /tmp/gnt_${v}_$grp -- python3 bench_train.py --device-input --no-oracle --no-scaling-model --steps 3 > /dev/null 2>&1
  done
  rm -rf /tmp/gnt_${v}_trace
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gnt_${v}_trace -- python3 bench_train.py --device-input --no-oracle --no-scaling-model --steps 3 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, collections, sys
v = sys.argv[1]
for d in sorted(glob.glob(f"/tmp/gnt_{v}_[A-Z]*")):
    fs = sorted(glob.glob(d + "/*/*_counter_collection.csv"))
    if not fs:
        print(d, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        for k in ("k_bucket_group5", "k_piece_part", "k_pair_rows", "k_piece_bounds"):
            if k in r["Kernel_Name"]:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), x in sorted(acc.items()):
        print(f"{k:18s} {c:12s} {sum(x) / len(x) / 1024:10.1f} MiB   (n={len(x)})")
for f in glob.glob(f"/tmp/gnt_{v}_trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("k_bucket_group5", "k_piece_part", "k_pair_rows", "k_piece_bounds")):
            print(r["Name"][:40], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
done
