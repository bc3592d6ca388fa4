#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of `yacht train` at configs[3], sketches in HBM, under tuning builds
# (lib/libyacht_hip_<name>.so) and tuning environments.  usage (GPU box, repo root):
#   bash scripts/sweep_train_variants.sh "name[:ENV=VAL[,ENV=VAL]] ..."        ("default" = lib/libyacht_hip.so)
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$PWD
export TMPDIR=/tmp
for spec in $1; do
    v=${spec%%:*}
    envs=""
    [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
    lib=$ROOT/yacht_amd/lib/libyacht_hip.so
    [ "$v" != default ] && lib=$ROOT/yacht_amd/lib/libyacht_hip_$v.so
    rm -rf /tmp/prof_v
    (cd /tmp && env YH_DEBUG_TUNING=1 YACHT_HIP_LIB=$lib $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o t -- python3 "$ROOT/bench_train.py" --device-input --no-oracle --no-scaling-model --steps 5 > /tmp/prof_v.json 2> /tmp/prof_v.err)
    python3 - "$spec" <<'PY'
import csv, glob, json, sys
spec = sys.argv[1]
try:
    d = json.loads(open("/tmp/prof_v.json").read().strip().splitlines()[-1])
    ok = d["full_size_vs_genuine_reference"]["all_equal"]
except Exception as ex:
    ok = "no line: " + repr(ex)[:80]
f = sorted(glob.glob("/tmp/prof_v/**/*kernel_stats.csv", recursive=True))
out = []
tot = 0.0
if f:
    for r in csv.DictReader(open(f[-1])):
        nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if nm.startswith("k_"):
            out.append("%s %.0f" % (nm, float(r["AverageNs"]) / 1e3))
            if nm.startswith(("k_piece", "k_part", "k_bucket", "k_pair_rows")):
                tot += float(r["AverageNs"]) / 1e3
print("%-40s golden %s  sum %.0f us | %s" % (spec, ok, tot, "  ".join(out[:6])))
PY
done
