# usage (GPU box): bash scripts/ablate_lookup.sh  -- what the hits cost the lookup: timing-only builds (results wrong) without the
# posting-list walk of shared hashes (abl1), without the flush of the LDS hit table = no global atomics (abl2), without both (abl3).
# Build them first:  python -c "from yacht_amd import build; [build.build_variant('abl%d' % v, {'YH_ABLATE_LOOKUP': v}) for v in (1, 2, 3)]"
B="timeout 200 python bench.py --no-batched --no-cpu-baseline --no-train --no-scaling-model --no-host-inclusive --no-real-shape --steps 300 --percentile-steps 300"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "fused ms/step", d["ms_per_step"], "plain lookup kernel ms", d["paths"]["indexed"]["lookup_kernel_ms_avg"], "plain step", d["paths"]["indexed"]["ms_per_step"])'
for v in base abl1 abl2 abl3; do
  if [ "$v" = base ]; then L=$PWD/yacht_amd/lib/libyacht_hip.so; else L=$PWD/yacht_amd/lib/libyacht_hip_$v.so; fi
  [ -f "$L" ] && YACHT_HIP_LIB=$L $B 2>/dev/null | python -c "$P" $v
done
