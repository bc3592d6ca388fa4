#!/bin/bash
# The batched lookup by hashes in flight per lane (YH_BATCH_U = 1, 2, 4, 8: build variants), two takes each.
cd "$GRAFT_REPO_ROOT" || exit 1
python3 - <<'PY'
from yacht_amd import build
for u in (1, 2, 4, 8):
    build.build_variant(f"bu{u}", {"YH_BATCH_U": u})
PY
for u in 1 2 4 8; do for take in 1 2; do
  YACHT_HIP_LIB="yacht_amd/lib/libyacht_hip_bu$u.so" python3 bench.py --no-train --no-sketch --no-cpu-baseline --no-host-inclusive --no-real-shape 2>/dev/null | tail -1 > /tmp/line.json
  python3 - "$u" <<'PY'
import json, sys
d = json.loads(open("/tmp/line.json").read())
sm = d["scaling_model"]["per_G"]
print(f"U={sys.argv[1]}: batched {d['batched']['ms_per_sample']:.4f} ms/sample (equal {d['batched']['equals_single_sample_step']})  rank-0 share per block at G=2/4/8: "
      + " / ".join(f"{sm[g]['batched_rank0_ms_per_block']:.3f}" for g in ('2', '4', '8')) + " ms", flush=True)
PY
done; done
