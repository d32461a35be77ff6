#!/bin/bash
# the round's committed profiles: config 2 (bench, kernel stats, categories, five PMC passes) and configs 3 / 4 / 5; usage (GPU box): bash tools/run_profiles.sh r04
TAG=${1:-r04}
rm -rf gpurun_out/prof_$TAG
bash tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
# (rocprofv3's PMC passes occasionally die inside the tool with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT: repeat the missing ones once)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$TAG
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass
  if [ ! -f $O/${TAG}_pmc_$1.json ]; then
    (cd /tmp; rm -rf $O/$1; timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -o p -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/$1.log 2>&1; python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_$1.json $O/$1/p_counter_collection.csv && rm -rf $O/$1)
  fi
done
bash tools/profile_configs.sh $TAG "3 4 5" > gpurun_out/profile_configs_$TAG.log 2>&1
timeout 300 python tools/perf_pfd.py 2>&1 | grep -v amdgpu > $O/${TAG}_perf_pfd.txt
ls $O | grep -c json
