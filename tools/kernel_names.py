"""Unique kernel names of a rocprofv3 kernel trace, with launch counts (full names: the input of tools/foreign_isa_scan.py --names).
  python tools/kernel_names.py <kernel_trace.csv> <out.txt> [tail_ms]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
if len(sys.argv) > 3:
    t_end = max(int(r["End_Timestamp"]) for r in rows)
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - float(sys.argv[3]) * 1e6]
cnt = collections.Counter(r["Kernel_Name"] for r in rows)
with open(sys.argv[2], "w") as f:
    for n, _ in cnt.most_common():
        f.write(n + "\n")
with open(sys.argv[2] + ".counts", "w") as f:
    for n, c in cnt.most_common():
        f.write("%7d  %s\n" % (c, n))
print("%d launches, %d distinct kernels -> %s" % (len(rows), len(cnt), sys.argv[2]))
