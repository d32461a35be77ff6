R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c72; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/ks -o k -- python3 $R/tools/step_only.py --config 2 --steps 12 --warmup 4 > $O/step_only.log 2>&1
grep STEP_ONLY $O/step_only.log
python3 $R/tools/gaps.py $O/ks/k_kernel_trace.csv 25 | tee $O/gaps_c2.txt
python3 $R/tools/busy.py $O/ks/k_kernel_trace.csv | tail -3
rm -rf $O/ks
