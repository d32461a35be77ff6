"""kernel-trace csv -> per-step table of every kernel above a floor.  usage: kconv.py trace.csv n_steps [floor_us]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1]))); nst = float(sys.argv[2]); floor = float(sys.argv[3]) if len(sys.argv) > 3 else 30
names = collections.defaultdict(lambda: [0, 0])
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    n = re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name']); n = re.sub(r"^void ", "", n).split("(")[0][:70]
    names[n][0] += 1; names[n][1] += d
tot = sum(v[1] for v in names.values())
print("total %.3f ms/step, %d launches/step" % (tot / nst / 1e6, sum(v[0] for v in names.values()) / nst))
acc = 0
for k, v in sorted(names.items(), key=lambda kv: -kv[1][1]):
    acc += v[1]
    if v[1] / nst < floor * 1e3: continue
    print("%7.1f x %8.1f us = %8.1f us  (cum %5.1f%%)  %s" % (v[0] / nst, v[1] / v[0] / 1e3, v[1] / nst / 1e3, 100 * acc / tot, k))
