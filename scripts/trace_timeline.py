#!/usr/bin/env python3
"""Print the merged kernel / memory-copy timeline of a rocprofv3 trace directory (csv): start offset, duration, what."""
import csv
import glob
import sys

d = sys.argv[1]
lo, hi = int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 80
rows = []
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K  " + r["Kernel_Name"][:50], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(d + "/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "CP " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), r.get("Stream_Id", "")))
rows.sort()
# keep the tail: the steady state of the loop
rows = rows[-hi:] if lo == 0 else rows[lo:hi]
t0 = rows[0][0]
for s, e, name, q in rows:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  {name}  [{q}]")
