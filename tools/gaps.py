"""The largest idle gaps of the LAST full step of a rocprofv3 kernel trace (steps delimited by adamw_multi_kernel): for each, how long no
kernel ran on any queue, the kernels that ended last before it and the one that started after it -- where the step waits for the host
(or for a cross-stream event) rather than for the GPU.   python tools/gaps.py <kernel_trace.csv> [n]"""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
marks = [e for s, e, k, q in rows if 'adamw_multi' in k]
a, b = marks[-2], marks[-1]
ks = [r for r in rows if r[0] >= a and r[1] <= b]
short = lambda k: k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
gaps, end, last = [], ks[0][1], ks[0]
for r in ks[1:]:
    if r[0] > end:
        gaps.append((r[0] - end, end - a, last, r))
    if r[1] > end:
        end, last = r[1], r
print("last step %.2f ms, %d kernels, idle %.2f ms in %d gaps (> 20 us: %.2f ms in %d)" % (
    (b - a) / 1e6, len(ks), sum(g[0] for g in gaps) / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 20000) / 1e6, sum(1 for g in gaps if g[0] > 20000)))
for g, at, p, q in sorted(gaps, reverse=True)[:n]:
    print("%7.1f us at +%6.2f ms   after %-60s (q %s)   before %-60s (q %s)" % (g / 1e3, at / 1e6, short(p[2]), p[3], short(q[2]), q[3]))
