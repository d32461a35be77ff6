"""Where a step's wall time goes, from a rocprofv3 kernel trace (steps delimited by the adamw_multi kernel):
busy / idle time, time with two or more kernels in flight, per-queue sums, and the launches bucketed by workgroup count
(a launch with fewer workgroups than the chip has CUs cannot fill it: its time is latency, not throughput).

  python tools/lanes.py <kernel_trace.csv> [steps]
"""
import csv
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from timeline import short


def _int(r, *names):
    for n in names:
        if n in r and r[n] not in ("", None):
            return int(r[n])
    return 0


def main(path, last=3):
    rows = []
    for r in csv.DictReader(open(path)):
        wg = max(1, _int(r, "Workgroup_Size_X", "Workgroup_Size")) * max(1, _int(r, "Workgroup_Size_Y")) * max(1, _int(r, "Workgroup_Size_Z"))
        grid = max(1, _int(r, "Grid_Size_X", "Grid_Size")) * max(1, _int(r, "Grid_Size_Y")) * max(1, _int(r, "Grid_Size_Z"))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), max(1, grid // wg)))
    rows.sort()
    marks = [e for s, e, n, q, g in rows if "adamw_multi" in n]
    for a, b in list(zip(marks, marks[1:]))[-last:]:
        ks = [k for k in rows if k[0] >= a and k[1] <= b]
        ev = sorted([(s, 1) for s, e, *_ in ks] + [(e, -1) for s, e, *_ in ks])
        depth, t0, at = 0, a, defaultdict(int)
        for t, d in ev:
            at[min(depth, 3)] += t - t0
            depth, t0 = depth + d, t
        at[0] += b - t0
        wall = b - a
        print("step %.2f ms, %d kernels: idle %.2f ms, one kernel in flight %.2f, two %.2f, three or more %.2f; kernel time %.2f ms"
              % (wall / 1e6, len(ks), at[0] / 1e6, at[1] / 1e6, at[2] / 1e6, at[3] / 1e6, sum(k[1] - k[0] for k in ks) / 1e6))
        q = defaultdict(lambda: [0, 0])
        for s, e, n, qi, g in ks:
            q[qi][0] += 1
            q[qi][1] += e - s
        print("  queues: " + ", ".join("%s: %d launches %.2f ms" % (k, v[0], v[1] / 1e6) for k, v in sorted(q.items(), key=lambda kv: -kv[1][1])))
        buckets = ((1, 16), (16, 64), (64, 256), (256, 1024), (1024, 1 << 40))
        for lo, hi in buckets:
            sel = [k for k in ks if lo <= k[4] < hi]
            if sel:
                print("  %5d <= workgroups < %-6s %5d launches %8.2f ms (%.1f us each)"
                      % (lo, hi if hi < (1 << 40) else "inf", len(sel), sum(k[1] - k[0] for k in sel) / 1e6, sum(k[1] - k[0] for k in sel) / 1e3 / len(sel)))
        # time each launch spends as the ONLY kernel in flight, by bucket and by kernel name
        alone_b, alone_n = defaultdict(int), defaultdict(lambda: [0, 0])
        evs = sorted([(k[0], 0, i) for i, k in enumerate(ks)] + [(k[1], -1, i) for i, k in enumerate(ks)])
        live, t0 = set(), a
        for t, kind, i in evs:
            if len(live) == 1 and t > t0:
                j = next(iter(live))
                g = ks[j][4]
                alone_b[next(lo for lo, hi in buckets if lo <= g < hi)] += t - t0
                nm = ks[j][2]
                alone_n[nm][0] += 1
                alone_n[nm][1] += t - t0
            if kind == 0:
                live.add(i)
            else:
                live.discard(i)
            t0 = t
        print("  alone in flight, by workgroup count: " + ", ".join("%d+: %.2f ms" % (lo, alone_b[lo] / 1e6) for lo, hi in buckets))
        for nm, (n, t) in sorted(alone_n.items(), key=lambda kv: -kv[1][1])[:14]:
            print("    alone %7.2f ms in %4d intervals  %s" % (t / 1e6, n, nm))
        gaps = []
        end = ks[0][1]
        for s, e, *_ in ks[1:]:
            if s > end:
                gaps.append(s - end)
            end = max(end, e)
        print("  %d idle gaps, %.2f ms in total, median %.1f us, %d above 10 us"
              % (len(gaps), sum(gaps) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0, sum(1 for g in gaps if g > 10000)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
