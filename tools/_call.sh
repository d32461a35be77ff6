R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c31; mkdir -p $O; cd $R
timeout 600 python3 tools/x6/ab_fwd_env.py DSF_X6P_NSLOW 0 1 > $O/ab_nslow.txt 2>&1; grep -v amdgpu.ids $O/ab_nslow.txt
timeout 600 python3 tools/x6/ab_fwd_env.py DSF_X6P_BD 2 3 > $O/ab_bd.txt 2>&1; grep -v amdgpu.ids $O/ab_bd.txt | head -4
