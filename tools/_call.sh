R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c55; mkdir -p $O; cd $R
timeout 600 python3 tools/pfd_pairs.py 5 6 2>&1 | grep -v amdgpu > $O/pfd_pairs_c5.txt; cat $O/pfd_pairs_c5.txt
timeout 600 python3 tools/pfd_pairs.py 3 6 2>&1 | grep -v amdgpu > $O/pfd_pairs_c3.txt; cat $O/pfd_pairs_c3.txt
