R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c82; mkdir -p $O; cd $R
for spec in "DSF_X6_WRW_WGS 0 320 384" "DSF_BN_VAR 1 3" "DSF_BN_BWD_WGS 1024 512 2048" "DSF_X6_KSPLIT_WGS 512 256 768"; do
  set -- $spec; var=$1; shift
  timeout 900 python3 tools/ab_env.py --config 2 --var $var --values "$@" --block 10 --rounds 8 2>&1 | grep "^AB" | tee -a $O/ab_tune.txt
done
