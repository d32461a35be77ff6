"""Per-queue view of one traced step (steps delimited by adamw_multi): for every hardware queue its kernel count, busy time and the
interval it covers; the phases of the step (forward / backward / tail) located by kernel names; and every interval > 15 us in which
the MAIN queue (the one with the most kernels) is idle, with what the other queues run meanwhile.
    python tools/queues.py <kernel_trace.csv> [step index from the end, default 1]"""
import sys
sys.path.insert(0, 'tools')
import timeline
rows = timeline.load(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
marks = [e for s, e, n, q, g in rows if "adamw_multi" in n]
a, b = marks[-1 - back], marks[-back]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
qs = {}
for s, e, n, q, g in sel:
    qs.setdefault(q, []).append((s, e, n, g))
main = max(qs, key=lambda q: len(qs[q]))
print("step %.1f us, %d kernels, %d queues; main queue q%s" % ((b - a) / 1e3, len(sel), len(qs), main))
for q, ks in sorted(qs.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, n, g in ks)
    print("  q%-3s %4d kernels, busy %8.1f us, from %8.1f to %8.1f   e.g. %s" % (q, len(ks), busy / 1e3, (ks[0][0] - a) / 1e3, (ks[-1][1] - a) / 1e3, ks[len(ks) // 2][2][:50]))
ks = qs[main]
tot = 0
print("main-queue idle intervals > 15 us:")
for (s0, e0, n0, g0), (s1, e1, n1, g1) in zip(ks, ks[1:]):
    gap = s1 - e0
    if gap > 15000:
        tot += gap
        others = []
        for q, kk in qs.items():
            if q == main: continue
            ov = sum(max(0, min(e, s1) - max(s, e0)) for s, e, n, g in kk)
            if ov > 0:
                names = sorted({n[:28] for s, e, n, g in kk if min(e, s1) - max(s, e0) > 0})
                others.append("q%s %.0f us (%s)" % (q, ov / 1e3, ", ".join(names[:3])))
        print("  %7.1f us at %8.1f  after %-40s before %-40s | %s" % (gap / 1e3, (e0 - a) / 1e3, n0[:40], n1[:40], "; ".join(others) or "GPU idle"))
print("main queue idle (intervals > 15 us): %.1f us; main queue busy %.1f us" % (tot / 1e3, sum(e - s for s, e, n, g in ks) / 1e3))
