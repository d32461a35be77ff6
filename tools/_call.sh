set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c10; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_steps.py tests/test_gpu_properties.py tests/test_gpu_step_ops.py tests/test_gpu_determinism.py -q -m gpu -x > $O/pytest_sub.log 2>&1; echo "rc=$?" >> $O/pytest_sub.log
tail -n 4 $O/pytest_sub.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/grid_barrier tools/platform/grid_barrier.hip && timeout 120 /tmp/grid_barrier > $O/grid_barrier.txt 2>&1
cat $O/grid_barrier.txt
timeout 600 python3 tools/ab_env.py --config 2 --var DSF_BN_PROBE --values none skip > $O/ab_probe_c2.txt 2>&1
timeout 900 python3 tools/ab_env.py --config 4 --var DSF_BN_PROBE --values none skip --block 5 --rounds 4 > $O/ab_probe_c4.txt 2>&1
timeout 900 python3 tools/ab_env.py --config 5 --var DSF_BN_PROBE --values none skip --block 5 --rounds 4 > $O/ab_probe_c5.txt 2>&1
grep AB $O/ab_probe_c*.txt
timeout 600 python3 tools/launch_sources.py --config 5 --top 5 2>&1 | grep "launches per step" | cut -c1-500
