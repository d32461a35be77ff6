"""Two-stream view of a traced training step: how long each HIP queue is busy, how much of that overlaps, and which queue finishes
the backward pass.   python tools/streams.py k_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
qcol = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r[qcol]) for r in rows)
marks = [e for s, e, n, q in ks if "adamw_multi" in n]
a, b = marks[-2], marks[-1]
step = [k for k in ks if k[0] >= a and k[1] <= b]
queues = sorted(set(k[3] for k in step), key=lambda q: -sum(k[1] - k[0] for k in step if k[3] == q))


def union(iv):
    tot, cs, ce = 0, None, None
    for s, e in sorted(iv):
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if ce is not None else 0)


print("step %.2f ms, %d kernels, queues %s" % ((b - a) / 1e6, len(step), queues))
busy = {q: union([(k[0], k[1]) for k in step if k[3] == q]) for q in queues}
allb = union([(k[0], k[1]) for k in step])
for q in queues:
    sel = [k for k in step if k[3] == q]
    print("  queue %s: %4d kernels, busy %6.2f ms, first at %6.2f ms, last ends at %6.2f ms (%s)" % (
        q, len(sel), busy[q] / 1e6, (sel[0][0] - a) / 1e6, (sel[-1][1] - a) / 1e6, sel[-1][2].replace("(anonymous namespace)::", "")[:40]))
print("  any queue busy %.2f ms; sum of queues %.2f ms -> %.2f ms of the step run two queues at once" % (
    allb / 1e6, sum(busy.values()) / 1e6, (sum(busy.values()) - allb) / 1e6))
if len(queues) > 1:
    main, side = queues[0], queues[1]
    s0 = min(k[0] for k in step if k[3] == side)
    s1 = max(k[1] for k in step if k[3] == side)
    m_in = union([(max(k[0], s0), min(k[1], s1)) for k in step if k[3] == main and k[1] > s0 and k[0] < s1])
    print("  second queue active window %.2f .. %.2f ms (%.2f ms): busy %.2f ms itself, main busy %.2f ms inside it" % (
        (s0 - a) / 1e6, (s1 - a) / 1e6, (s1 - s0) / 1e6, busy[side] / 1e6, m_in / 1e6))
    after = [k for k in step if k[3] == main and k[0] >= s1]
    print("  main-queue kernels after the second queue's last kernel: %d (%.2f ms)" % (len(after), sum(k[1] - k[0] for k in after) / 1e6))
