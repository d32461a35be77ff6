R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c52; mkdir -p $O; cd $R
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_BN_WRITE_G --values 1 0 --rounds 6 > $O/ab_writeg.txt 2>&1; tail -n 2 $O/ab_writeg.txt
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_BN_TWIN --values 1 0 --rounds 6 > $O/ab_twin.txt 2>&1; tail -n 2 $O/ab_twin.txt
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_WRW_BIAS_MAX_SPLITS --values 64 0 256 --rounds 6 > $O/ab_biasms.txt 2>&1; tail -n 3 $O/ab_biasms.txt
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_X6_WRW_PATCH --values 1 0 --rounds 6 > $O/ab_wrwpatch.txt 2>&1; tail -n 2 $O/ab_wrwpatch.txt
timeout 900 python3 tools/ab_env.py --config 5 --var DSF_CROP_WG_TARGET --values 0 512 2048 --rounds 3 --block 5 > $O/ab_crop.txt 2>&1; tail -n 3 $O/ab_crop.txt
