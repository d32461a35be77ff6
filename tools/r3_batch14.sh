#!/bin/bash
# round 3, final state: profile round at HEAD (BatchNorm kernels changed since the last set), other configs, soak, smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b14; mkdir -p $O
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1
bash tools/profile_configs.sh r03 > $O/profile_configs.log 2>&1
timeout 600 python3 tools/soak.py > $O/soak.log 2>&1; tail -3 $O/soak.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"
tail -2 $O/smoke.log
