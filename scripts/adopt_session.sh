#!/bin/bash
# HERE (not on the GPU box), after `gpurun -- bash scripts/measure_session.sh`: only gpurun_out/ travels back, so the traffic files
# bench.py / bench_train.py attach (keyed on the kernels' source) and the evidence the documents cite are copied into profiles/.
# usage: bash scripts/adopt_session.sh [round tag, default r06]
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
for n in fused index stream train; do
  [ -f gpurun_out/traffic_${TAG}_$n.json ] && cp gpurun_out/traffic_${TAG}_$n.json profiles/traffic_${TAG}_$n.json
done
mkdir -p profiles/$TAG
for f in bench_n1.json bench_train.json train_kernel_stats.txt train_kernel_stats_device_input.txt pmc_train_kernels.txt summary.txt traffic.txt \
         kernel_stats.csv pmc_fetch.csv pmc_sq1.csv pmc_sq2.csv pmc_tcc.csv pmc_write.csv e2e_run.json e2e_train.json train_share_probe.txt \
         tests_gpu.txt fuzz_campaign.txt; do
  [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/$TAG/$f
done
python3 - "$TAG" <<'PY'
import hashlib, json, sys
tag = sys.argv[1]
def src(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open("yacht_amd/csrc/" + f, "rb").read())
    return h.hexdigest()[:16]
now = {"run": src(("yh_query.hip", "yh_common.h")), "train": src(("yh_sort.hip", "yh_pairwise.hip", "yh_common.h"))}
for n in ("fused", "index", "stream", "train"):
    t = json.load(open(f"profiles/traffic_{tag}_{n}.json"))
    want = now["train" if n == "train" else "run"]
    print(f"profiles/traffic_{tag}_{n}.json  source_tag {t['source_tag']}  {'matches this tree' if t['source_tag'] == want else 'STALE: the kernels changed since (' + want + ')'}")
PY
