R=$GRAFT_REPO_ROOT; cd $R
bash tools/run_profiles.sh r06
python3 -c "import json; d=json.loads(open('gpurun_out/prof_r06/r06_bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
