#!/bin/bash
# BN reduce grid sweep (GPU box): builds a probe library (tools/probes/build_probe.sh -DDSF_BN_MAX_WGS=n), prints the BN kernels' per-step time from a kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  bash $R/tools/probes/build_probe.sh -DDSF_BN_MAX_WGS=$n > /tmp/build_$n.log 2>&1 || { tail -5 /tmp/build_$n.log; continue; }
  rm -rf /tmp/ks; rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -o k -- python3 $R/tools/step_only.py > /dev/null 2>&1
  echo "== BN_MAX_WGS $n"; python3 $R/tools/kconv.py /tmp/ks/k_kernel_trace.csv 12 100 | grep "total\|bn_"
done
bash $R/tools/probes/build_probe.sh --restore > /dev/null
