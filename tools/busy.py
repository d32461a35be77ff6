"""GPU busy fraction per step from a rocprofv3 kernel trace: steps are delimited by the adamw_multi kernel."""
import csv, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
marks = [e for s, e, n in rows if 'adamw_multi' in n]
for a, b in list(zip(marks, marks[1:]))[-6:]:
    ks = [(s, e) for s, e, n in rows if s >= a and e <= b]
    busy, cur_s, cur_e = 0, None, None
    for s, e in ks:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = sorted(((ks[i + 1][0] - max(k[1] for k in ks[:i + 1][-8:])) for i in range(len(ks) - 1)), reverse=True)
    print(f"step {(b-a)/1e6:.2f} ms, kernels {len(ks)}, busy {busy/1e6:.2f} ms ({100*busy/(b-a):.1f}%), sum {sum(e-s for s,e in ks)/1e6:.2f} ms, largest gaps us: {[round(g/1e3,1) for g in gaps[:6]]}")
