cd $GRAFT_REPO_ROOT
cp dsf_amd/lib/libdsf_hip.so /tmp/new.so; cp dsf_amd/lib/libdsf_hip_ex.so /tmp/old.so
for rep in 1 2 3; do for v in new old; do cp /tmp/$v.so dsf_amd/lib/libdsf_hip.so
  for c in 5 3; do python bench.py --config $c --steps 16 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c fresh', 'culled-only' if '$v'=='new' else 'with-exhaustive-role', d['ms_per_step'], 'ms')"; done
done; done
cp /tmp/new.so dsf_amd/lib/libdsf_hip.so
