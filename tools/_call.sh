R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c59; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_edge.py tests/test_gpu_stress.py tests/test_gpu_steps.py -q -m gpu > $O/pytest_geo.log 2>&1; echo "rc=$?" >> $O/pytest_geo.log; tail -n 2 $O/pytest_geo.log
timeout 300 python3 tools/perf_pfd.py 2>&1 | grep -v amdgpu | grep "B 64\|labelled" > $O/perf_pfd.txt; cat $O/perf_pfd.txt
timeout 600 python3 tools/pfd_in_step.py 5 3 2>&1 | grep -v amdgpu > $O/pfd_in_step.txt; cat $O/pfd_in_step.txt
cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 $R/tools/step_only.py --config 5 --steps 8 --warmup 3 > /dev/null 2>&1; grep -E "mesh_point_fwd" $O/kt/k_kernel_stats.csv | sed 's/.*)",//'; rm -rf $O/kt
