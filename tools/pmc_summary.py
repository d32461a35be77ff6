"""Aggregates rocprofv3 `--pmc ... --output-format csv` counter_collection files per kernel name.
usage: pmc_summary.py out.json file1.csv [file2.csv ...]   -> {kernel: {counter: {"sum":, "avg":, "n":}}}"""
import csv, json, sys, collections, re

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:120]

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for path in sys.argv[2:]:
    seen = set()
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            a = agg[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
            key = (path, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                d = dur[k]; d[0] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"]); d[1] += 1
out = {}
for k, cs in agg.items():
    out[k] = {c: {"sum": v[0], "avg": v[0] / v[1], "n": v[1]} for c, v in cs.items()}
    out[k]["_duration_ns_under_pmc"] = {"sum": dur[k][0], "avg": dur[k][0] / max(dur[k][1], 1), "n": dur[k][1]}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
