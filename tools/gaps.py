"""Where a training step's GPU-idle time sits: rocprofv3 kernel trace -> idle intervals (no kernel of any stream running) of the last
step, by size class, and the kernels on both sides of the largest ones.   python tools/gaps.py k_kernel_trace.csv"""
import csv
import sys

rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
marks = [e for s, e, n in rows if 'adamw_multi' in n]
a, b = marks[-2], marks[-1]
ks = [(s, e, n) for s, e, n in rows if s >= a and e <= b]
gaps, reach, last = [], ks[0][1], ks[0][2]
for s, e, n in ks[1:]:
    if s > reach:
        gaps.append((s - reach, (reach - a) / 1e6, last, n))
    if e > reach:
        reach, last = e, n
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')[:60]
print("step %.2f ms, %d kernels, idle %.3f ms in %d gaps" % ((b - a) / 1e6, len(ks), sum(g[0] for g in gaps) / 1e6, len(gaps)))
for lo, hi in ((0, 1000), (1000, 2000), (2000, 4000), (4000, 10000), (10000, 10 ** 9)):
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print("  gaps %5.1f-%-6s us: %4d, %7.1f us" % (lo / 1e3, "%.1f" % (hi / 1e3) if hi < 10 ** 9 else "inf", len(sel), sum(sel) / 1e3))
# idle by phase: forward = until the first backward kernel (bn_bwd / huber_bwd), backward = rest
tb = next((s for s, e, n in ks if 'bwd' in n or 'backward' in n), b)
print("  idle before the first backward kernel (%.2f ms into the step): %.1f us; after: %.1f us" % (
    (tb - a) / 1e6, sum(g[0] for g in gaps if a + g[1] * 1e6 < tb) / 1e3, sum(g[0] for g in gaps if a + g[1] * 1e6 >= tb) / 1e3))
for g in sorted(gaps, reverse=True)[:12]:
    print("  %6.1f us at %6.2f ms: %s  ->  %s" % (g[0] / 1e3, g[1], short(g[2]), short(g[3])))
