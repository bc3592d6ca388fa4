#!/bin/bash
# What each part of the one-launch step (k_step_fused) costs: timing-only builds (results wrong) that compile parts out, one
# after the other, down to "read the sample and nothing else".  VERDICT r04 "next" 4: the ~5-7 us between the access pattern
# alone (scripts/probes/gather_probe3: 27.0 us) and the kernel (34.3 us).
# Build first (build container):  for v in 32 64 3 7 15 31 127: build.build_variant('l%d' % v, {'YH_ABLATE_LOOKUP': v}); 'lb': {'YH_LOOKUP_LATE_BAD': 1}
# usage (GPU box, repo root): bash scripts/ablate_step.sh
cd "$GRAFT_REPO_ROOT" || exit 1
B="timeout 300 python bench.py --no-batched --no-cpu-baseline --no-train --no-scaling-model --no-host-inclusive --no-real-shape --steps 300 --percentile-steps 300 --min-timed-steps 4000"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%-8s %-58s fused step %.4f ms   kernel (events) %.1f us  net of event pair %.1f us   plain lookup kernel %.1f us" % (sys.argv[1], sys.argv[2], d["ms_per_step"], 1e3*r["kernel_ms_avg"], 1e3*r.get("kernel_ms_avg_net_of_event_overhead", 0), 1e3*d["paths"]["indexed"]["lookup_kernel_ms_avg"]))'
run() {
  v=$1; what=$2
  if [ "$v" = base ]; then L=$PWD/yacht_amd/lib/libyacht_hip.so; else L=$PWD/yacht_amd/lib/libyacht_hip_$v.so; fi
  [ -f "$L" ] && YACHT_HIP_LIB=$L $B 2>/dev/null | python -c "$P" "$v" "$what"
}
run base "the step as shipped"
run lb   "ordering verdict asked for BEHIND the sample's loads"
run l32  "no ordering verdict read at all"
run l64  "reducer + exclusive roles of the launch do nothing"
run l3   "no posting-list walk, no flush of the hit table"
run l7   "... and no hit is counted (no LDS atomics)"
run l15  "... and no bucket is read (filter word only)"
run l31  "... and no filter word either (sample read + barriers)"
run l127 "... and no verdict, no tail roles: the launch itself"
run base "the step as shipped (again)"
