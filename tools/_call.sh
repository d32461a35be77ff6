R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c53; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "whole_network_weight_split" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 5 $O/pytest.log
timeout 300 python3 tools/step_only.py --config 2 --steps 20 --warmup 5 2>&1 | grep STEP_ONLY
cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 $R/tools/step_only.py --config 2 --steps 6 --warmup 3 > /dev/null 2>&1; grep -E "split_weights_multi|adamw_multi" $O/kt/k_kernel_stats.csv | cut -c1-160; rm -rf $O/kt
