"""kernel-trace csv -> per-step category table.  usage: kstats.py kernel_trace.csv n_steps [tail_ms]
With tail_ms only the kernels that START within the last tail_ms milliseconds of the trace are counted (the timed steps of
tools/step_only.py: warm-up, graph capture and validation passes fall away), divided by n_steps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nst = float(sys.argv[2])
if len(sys.argv) > 3:
    t_end = max(int(r['End_Timestamp']) for r in rows)
    rows = [r for r in rows if int(r['Start_Timestamp']) >= t_end - float(sys.argv[3]) * 1e6]
def c(n):
    if 'igemm' in n: return 'conv'
    if 'BatchNorm' in n or n.startswith('void (anonymous namespace)::bn_') or '::bn_' in n: return 'bn'
    if 'fillBuffer' in n or 'FillFunctor' in n: return 'fill'
    if 'copyBuffer' in n or 'CatArray' in n or 'direct_copy' in n: return 'copy'
    if 'multi_tensor' in n: return 'optimizer'
    if 'elementwise' in n: return 'elementwise'
    if 'reduce' in n.lower(): return 'reduce'
    if any(k in n for k in ('render','raster','mano','pfd','joint2offset','offset2joint','crop_','uvd','xyz','collision','bbox','project','huber','col_sum')): return 'dsf'
    return 'other'
agg = collections.defaultdict(lambda: [0, 0]); names = collections.defaultdict(lambda: [0, 0])
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = c(r['Kernel_Name']); agg[k][0] += 1; agg[k][1] += d
    names[r['Kernel_Name'][:150]][0] += 1; names[r['Kernel_Name'][:150]][1] += d
tot = sum(v[1] for v in agg.values())
print(f"kernels total {tot/1e6:.1f} ms over {(t1-t0)/1e6:.1f} ms wall; per step (/{nst:g}):")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:12s} {v[0]/nst:8.1f} launches  {v[1]/nst/1e6:7.3f} ms")
print("top non-conv kernels per step:")
for k, v in sorted(names.items(), key=lambda kv: -kv[1][1]):
    if 'igemm' in k: continue
    if v[1] / nst < 40e3: break
    print(f"  {v[0]/nst:7.1f} x {v[1]/v[0]/1e3:7.1f} us = {v[1]/nst/1e3:8.1f} us  {k[:130]}")
