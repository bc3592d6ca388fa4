#!/usr/bin/env python3
"""Per-block GPU time of the batched hash-range runner in steady state, from a rocprofv3 kernel trace of
scripts/probes/batch_share_trace.py: everything from the 4th k_batch_lookup on (the three warm-up blocks and the set-up dropped),
summed per kernel name and divided by the blocks left.   usage: batch_share_blocks.py <dir with *_kernel_trace.csv> [warm-up blocks]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
rows.sort()
look = [t for t, _, k in rows if "k_batch_lookup" in k]
if len(look) <= warm:
    sys.exit("no steady-state blocks in the trace")
t0 = look[warm]
blocks = len(look) - warm
agg, cnt = defaultdict(int), defaultdict(int)
busy, last_end = 0, t0
for s, e, k in rows:
    if s < t0:
        continue
    agg[k] += e - s
    cnt[k] += 1
    if e > last_end:  # (the union of the kernels' intervals: what the GPU was busy for)
        busy += e - max(s, last_end)
        last_end = e
wall = max(e for _, e, _ in rows) - t0
print("blocks %d   wall %.1f us per block   GPU busy %.1f us per block   sum of kernels %.1f us per block"
      % (blocks, wall / blocks / 1e3, busy / blocks / 1e3, sum(agg.values()) / blocks / 1e3))
for k in sorted(agg, key=lambda k: -agg[k]):
    print("%9.2f us per block  %6.2f launches per block  %8.2f us each   %s" % (agg[k] / blocks / 1e3, cnt[k] / blocks, agg[k] / cnt[k] / 1e3, k[:90]))
