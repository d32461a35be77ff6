set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 python3 $R/tools/graph_replay_ab.py > $O/graph_replay_ab.txt 2>&1
for m in forked single; do
  rm -rf $O/kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python3 $R/tools/graph_replay_ab.py --trace-mode $m > $O/trace_$m.log 2>&1
  python3 $R/tools/lanes.py $O/kt/k_kernel_trace.csv 4 > $O/lanes_graph_$m.txt 2>&1
  cp $O/kt/k_kernel_trace.csv $O/trace_$m.csv; rm -rf $O/kt
done
timeout 600 python3 $R/tools/ab_env.py --config 2 --var DSF_BN_VAR --values 0 2 3 7 > $O/ab_var_c2.txt 2>&1
timeout 900 python3 $R/tools/ab_env.py --config 4 --var DSF_BN_VAR --values 0 2 3 7 --block 5 --rounds 4 > $O/ab_var_c4.txt 2>&1
timeout 900 python3 $R/tools/ab_env.py --config 5 --var DSF_BN_VAR --values 0 2 3 7 --block 5 --rounds 4 > $O/ab_var_c5.txt 2>&1
for c in 2 3 4 5; do
  rm -rf $O/kc; timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/kc -o k -- python3 $R/tools/step_only.py --config $c --steps 4 --warmup 3 > $O/step_only_$c.log 2>&1
  python3 $R/tools/kernel_names.py $O/kc/k_kernel_trace.csv $O/kernel_names_config$c.txt
  rm -rf $O/kc
done
gzip -f $O/trace_forked.csv $O/trace_single.csv
ls -la $O
