#!/bin/bash
# The lookup by presence-filter form (GPU box): bits per hash inside the word (YH_FILTER_K = 1, 2, 3: build variants) x bits per
# distinct hash (YH_FILTER_BPH, behind the tuning gate), three takes each.  usage: bash scripts/sweep_filter.sh > gpurun_out/r04/filter_forms.txt
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
from yacht_amd import build
for k in (1, 2, 3):
    build.build_variant(f"fk{k}", {"YH_FILTER_K": k})
PY
run() {  # label, lib, bph
  for take in 1 2 3; do
  YACHT_HIP_LIB="$2" YH_DEBUG_TUNING=1 YH_FILTER_BPH="$3" python3 bench.py --no-train --no-sketch --no-scaling-model --no-cpu-baseline --no-host-inclusive --no-real-shape --min-timed-steps 4000 2>/dev/null | tail -1 > /tmp/line.json
  python3 - "$1" "$3" <<'PY'
import json, sys
d = json.loads(open("/tmp/line.json").read())
print(f"{sys.argv[1]:>5s} bits/hash {sys.argv[2]:>2s}: step {d['ms_per_step']:.4f} ms  kernel {d['roofline']['kernel_ms_avg']:.4f} ms  batched {d['batched']['ms_per_sample']:.4f} ms/sample  "
      f"filter {d['config']['filter_bytes'] / 1e6:.0f} MB  paths equal {all(p['equals_default_path'] for p in d['paths'].values())}  batched equal {d['batched']['equals_single_sample_step']}", flush=True)
PY
  done
}
L=yacht_amd/lib
run "k=1" "$L/libyacht_hip_fk1.so" 4
for bph in 3 4; do run "k=2" "$L/libyacht_hip_fk2.so" $bph; done
for bph in 3 4 5; do run "k=3" "$L/libyacht_hip_fk3.so" $bph; done
