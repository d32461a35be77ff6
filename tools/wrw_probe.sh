#!/bin/bash
# Timing probes of igemm_wrw_x6_kernel (GPU box): builds a probe library (tools/probes/build_probe.sh -DDSF_WRW_PROBE=n) and prints the kernel's launch time
# from bench.py's live replay.  0 = the shipped kernel.  Results are numerically WRONG for n > 0 (timing only).
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in 0 1 2 3 "$@"; do
  bash $R/tools/probes/build_probe.sh -DDSF_WRW_PROBE=$n > /tmp/build_$n.log 2>&1 || { tail -5 /tmp/build_$n.log; continue; }
  python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 4 2>/dev/null | python3 -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
ck = j['conv_kernels']
print('probe $n: step %.3f ms' % j['ms_per_step'], ' | '.join('%s %.1f us' % (k.replace('igemm_', ''), v['avg_launch_us']) for k, v in ck.items() if 'wrw_x6' in k))"
done
bash $R/tools/probes/build_probe.sh --restore > /dev/null
