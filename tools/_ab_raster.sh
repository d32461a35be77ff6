cd $GRAFT_REPO_ROOT
cp dsf_amd/lib/libdsf_hip.so /tmp/new.so; cp dsf_amd/lib/libdsf_hip_oldraster.so /tmp/old.so
for rep in 1 2 3; do for v in new old; do cp /tmp/$v.so dsf_amd/lib/libdsf_hip.so
  python bench.py --steps 20 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2 raster=$v', d['ms_per_step'], 'ms', d['roofline_raster']['avg_launch_us'], 'us alone')"
done; done
for rep in 1 2; do for v in new old; do cp /tmp/$v.so dsf_amd/lib/libdsf_hip.so
  for c in 5 3; do python bench.py --config $c --steps 10 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c raster=$v', d['ms_per_step'], 'ms')"; done
done; done
cp /tmp/new.so dsf_amd/lib/libdsf_hip.so
