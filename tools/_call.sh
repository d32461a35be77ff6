R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c38; mkdir -p $O; cd $R
timeout 600 python3 tools/x6/ab_fwd_env.py DSF_X6_PATCH 4 2 2>&1 | grep -v amdgpu.ids | head -6 > $O/ab_ip_layers.txt; cat $O/ab_ip_layers.txt
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_X6_PATCH --values 4 2 --rounds 8 > $O/ab_ip_c2.txt 2>&1; tail -n 2 $O/ab_ip_c2.txt
timeout 900 python3 tools/ab_env.py --config 5 --var DSF_X6_PATCH --values 4 2 --rounds 4 --block 5 > $O/ab_ip_c5.txt 2>&1; tail -n 2 $O/ab_ip_c5.txt
timeout 900 python3 tools/ab_env.py --config 4 --var DSF_X6_PATCH --values 4 2 --rounds 3 --block 4 > $O/ab_ip_c4.txt 2>&1; tail -n 2 $O/ab_ip_c4.txt
