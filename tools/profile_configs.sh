#!/bin/bash
# Per BASELINE config (bench.py --config 2|3|4|5): bench line, per-step kernel category table over the TIMED steps only
# (tools/step_only.py + tools/kstats.py with a time window), and the two PMC traffic passes (FETCH_SIZE, WRITE_SIZE; separate runs).
# usage (GPU box): bash tools/profile_configs.sh r04 "3 4 5"
TAG=${1:-r04}; CONFIGS=${2:-"3 4 5"}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in $CONFIGS; do
  timeout 900 python3 $R/bench.py --config $c > $O/${TAG}_bench_config$c.json 2> $O/bench_config$c.err
  rm -rf $O/kc; timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/kc -o k -- python3 $R/tools/step_only.py --config $c --steps 8 --warmup 3 > $O/step_only_$c.log 2>&1
  ms=$(grep STEP_ONLY $O/step_only_$c.log | sed 's/.*wall_ms \([0-9.]*\).*/\1/')
  if [ -z "$ms" ]; then echo "config $c: step_only.py printed no STEP_ONLY line (see $O/step_only_$c.log): skipped" >&2; continue; fi
  ( echo "config $c, per step over the 8 timed steps of tools/step_only.py (kernels starting in the last $ms ms of the trace; under the tracer a step takes $(grep STEP_ONLY $O/step_only_$c.log | sed 's/.*ms_per_step //') ms):"; python3 $R/tools/kstats.py $O/kc/k_kernel_trace.csv 8 $ms ) > $O/${TAG}_config${c}_kernel_categories.txt
  python3 $R/tools/lanes.py $O/kc/k_kernel_trace.csv 2 > $O/${TAG}_config${c}_lanes.txt 2>&1
  rm -rf $O/kc
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pm; timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pm -o p -- python3 $R/bench.py --config $c --steps 2 --warmup 2 --no-cpu-baseline > $O/pmc_${c}_${ctr}.log 2>&1
    python3 $R/tools/pmc_summary.py $O/${TAG}_config${c}_pmc_$ctr.json $O/pm/p_counter_collection.csv; rm -rf $O/pm
  done
  python3 - <<PY
import json
f, w = json.load(open("$O/${TAG}_config${c}_pmc_FETCH_SIZE.json")), json.load(open("$O/${TAG}_config${c}_pmc_WRITE_SIZE.json"))
out = {"source": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of bench.py --config $c --steps 2 --warmup 2 --no-cpu-baseline (tools/profile_configs.sh); KiB units, FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)", "kernels": {}}
for k in f:
    if "FETCH_SIZE" not in f[k]:
        continue
    fb = f[k]["FETCH_SIZE"]["avg"] * 1024 * 2
    wb = w.get(k, {}).get("WRITE_SIZE", {}).get("avg", 0.0) * 1024
    out["kernels"][k] = {"fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb, "launches_sampled": int(f[k]["FETCH_SIZE"]["n"])}
json.dump(out, open("$O/${TAG}_config${c}_pmc_traffic.json", "w"), indent=1, sort_keys=True)
PY
done
ls -la $O
