R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c88; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/ks -o k -- python3 $R/tools/step_only.py --config 2 --steps 12 --warmup 4 > $O/step_only.log 2>&1
head -1 $O/ks/k_kernel_trace.csv
python3 $R/tools/small_grids.py $O/ks/k_kernel_trace.csv 12 | tee $O/small_grids.txt
rm -rf $O/ks
