#!/bin/bash
# Bench line + rocprofv3 kernel stats of the other BASELINE configs (bench.py --config 3|4|5).  usage (GPU box): bash tools/profile_configs.sh r03
TAG=${1:-r03}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in 3 4 5; do
  timeout 900 python3 $R/bench.py --config $c > $O/${TAG}_bench_config$c.json 2> $O/bench_config$c.err
  rm -rf $O/kc; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kc -o k -- python3 $R/bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_config${c}_under_rocprof.json 2>> $O/bench_config$c.err
  python3 - <<PY > $O/${TAG}_config${c}_kernel_stats.txt
import csv, json
rows = list(csv.DictReader(open("$O/kc/k_kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
j = json.loads(open("$O/${TAG}_bench_config$c.json").read().strip().splitlines()[-1])
print("config $c: %s" % j["config"]["workload"])
print("bench line: %.2f %s, %.3f ms per step; roofline %s frac %.4f; cpu_baseline %s" % (j["value"], j["unit"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["frac"], json.dumps(j.get("cpu_baseline"))))
print("rocprofv3 --kernel-trace --stats of bench.py --config $c --steps 6 --warmup 3 --no-cpu-baseline (whole process: warm-up, timed steps, roofline replays), top 40 of %d kernels, %.1f ms of kernel time:" % (len(rows), tot / 1e6))
for r in rows[:40]:
    print("%6d x %9.1f us = %8.1f ms (%4.1f%%)  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:120]))
PY
  rm -rf $O/kc
done
ls -la $O
