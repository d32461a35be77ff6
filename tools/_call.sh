R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c96; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 3 $O/pytest_gpu.log
rm -rf $R/gpurun_out/prof_r06; mkdir -p $R/gpurun_out/prof_r06
bash tools/profile_configs.sh r06 "5" > gpurun_out/profile_configs_r06.log 2>&1
python3 -c "import json; d=json.loads(open('gpurun_out/prof_r06/r06_bench_config5.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
