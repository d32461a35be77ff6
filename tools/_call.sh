R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof_r06; mkdir -p gpurun_out/prof_r06
bash tools/profile_configs.sh r06 "5" > gpurun_out/profile_configs_r06.log 2>&1
ls gpurun_out/prof_r06
python3 -c "import json; d=json.loads(open('gpurun_out/prof_r06/r06_bench_config5.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
