"""One step of a rocprofv3 kernel trace as a text timeline (start us, duration us, queue, workgroups, kernel), steps
delimited by the adamw_multi kernel:  python tools/timeline.py <kernel_trace.csv> [step index from the end, default 1]"""
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"at::native::", "", name)
    return name.split("(")[0][:70]


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        w = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], max(1, g // w)))
    rows.sort()
    return rows


if __name__ == "__main__":
    rows = load(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    marks = [e for s, e, n, q, g in rows if "adamw_multi" in n]
    a, b = marks[-1 - back], marks[-back]
    for s, e, n, q, g in rows:
        if s >= a and e <= b:
            print("%9.1f %7.1f q%s wg%-6d %s" % ((s - a) / 1e3, (e - s) / 1e3, q, g, n))
