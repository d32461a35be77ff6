set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c7; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_step_ops.py tests/test_gpu_bn_pair.py -q -m gpu > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log
tail -n 30 $O/pytest_new.log
timeout 600 python3 tools/op_sources.py --config 5 --top 140 > $O/op_sources_c5.txt 2>&1
timeout 600 python3 tools/op_sources.py --config 2 --top 100 > $O/op_sources_c2.txt 2>&1
timeout 600 python3 tools/launch_sources.py --config 5 --top 60 > $O/launch_sources_c5.txt 2>&1
timeout 600 python3 tools/launch_sources.py --config 2 --top 60 > $O/launch_sources_c2.txt 2>&1
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -n 8 $O/pytest_gpu.log
timeout 600 python3 bench.py --config 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 python3 bench.py --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
cut -c1-260 $O/bench_c5.json $O/bench_c2.json
