#!/bin/bash
# rocprofv3 kernel stats of the other BASELINE configs' per-GPU steps (tools/other_configs.py).  usage (GPU box): bash tools/profile_other.sh r02
TAG=${1:-r02}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 python3 $R/tools/other_configs.py > $O/${TAG}_other_configs.txt 2> $O/other.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ko -o k -- python3 $R/tools/other_configs.py > /dev/null 2>> $O/other.err
( echo; echo "rocprofv3 --kernel-trace --stats of the same command (top 40 kernels by total time):"; python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/ko/k_kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print("%6d x %9.1f us = %8.1f ms (%4.1f%%)  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:110]))
PY
) >> $O/${TAG}_other_configs.txt
rm -rf $O/ko
