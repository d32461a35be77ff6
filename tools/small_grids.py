"""Kernels of the LAST full step of a rocprofv3 kernel trace that run with fewer workgroups than the chip has CUs (256) and take more than
a few microseconds, per queue: candidates for running beside something else (the pooled MANO head in front of the decoder was one:
65 us at 32 workgroups).   python tools/small_grids.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
marks = [int(r['End_Timestamp']) for r in rows if 'adamw_multi' in r['Kernel_Name']]
a, b = sorted(marks)[-2:]
def wgs(r):
    g = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    w = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
    return g // max(w, 1)
sel = [r for r in rows if int(r['Start_Timestamp']) >= a and int(r['End_Timestamp']) <= b]
sel.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda k: k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:64]
tot = {}
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = wgs(r)
    if n < 256 and d >= min_us:
        # how much of this kernel's span is covered by kernels of OTHER queues?
        s0, e0 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        ov = 0
        for o in sel:
            if o['Queue_Id'] != r['Queue_Id']:
                s1, e1 = int(o['Start_Timestamp']), int(o['End_Timestamp'])
                ov = max(ov, min(e0, e1) - max(s0, s1))
        print("+%6.2f ms  q%s  %4d workgroups  %7.1f us  (largest overlap with another queue's kernel %5.1f us)  %s" % (
            (s0 - a) / 1e6, r['Queue_Id'], n, d, max(ov, 0) / 1e3, short(r['Kernel_Name'])))
        k = (r['Queue_Id'])
        tot[k] = tot.get(k, 0) + d
print({("queue %s" % k): "%.0f us" % v for k, v in tot.items()})
