R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c84; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 3 $O/pytest_gpu.log
bash tools/run_profiles.sh r06
timeout 300 python3 tools/bn_pool_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r06/r06_stem_probe.txt
python3 -c "import json; d=json.loads(open('gpurun_out/prof_r06/r06_bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
