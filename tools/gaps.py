import sys
sys.path.insert(0, 'tools')
import timeline
rows = timeline.load(sys.argv[1])
marks = [e for s, e, n, q, g in rows if "adamw_multi" in n]
a, b = marks[-2], marks[-1]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
# union busy; list gaps > 20us with neighbours
cur_end = a
gaps = []
prev = None
for s, e, n, q, g in sel:
    if s > cur_end + 20000:
        gaps.append((s - cur_end, (cur_end - a) / 1e3, prev, n))
    if e > cur_end:
        cur_end = e; prev = n
print("step %.1f us, %d kernels" % ((b - a) / 1e3, len(sel)))
tot = 0
for d, at, p, n in sorted(gaps, reverse=True)[:25]:
    tot += d
    print("gap %7.1f us at %8.1f  after %-50s before %s" % (d / 1e3, at, p[:50], n[:60]))
print("sum of listed gaps %.1f us; all gaps>20us: %.1f us (%d)" % (tot / 1e3, sum(g[0] for g in gaps) / 1e3, len(gaps)))
