cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/ht; rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/ht -- python3 scripts/probes/train_trace_device.py > gpurun_out/hiptrace_train.err 2>&1
python3 - <<'PY'
import csv, glob
api = []
for f in glob.glob("/tmp/ht/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
kern = []
for f in glob.glob("/tmp/ht/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]))
api.sort(); kern.sort()
# the last pass: from the last hipStreamCreateWithFlags on
starts = [s for s, e, f in api if f == "hipStreamCreateWithFlags"]
t0 = starts[-1]
out = open("gpurun_out/hiptrace_train.txt", "w")
ev = [(s, e, "API " + f) for s, e, f in api if s >= t0] + [(s, e, "   KERNEL " + k) for s, e, k in kern if s >= t0]
ev.sort()
for s, e, n in ev[:400]:
    out.write("%9.1f us  +%7.1f us  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, n))
out.close()
PY
