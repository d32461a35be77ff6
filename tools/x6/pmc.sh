#!/bin/bash
# PMC view of igemm_x6_kernel on one shape: bash tools/x6/pmc.sh 32 64 488 256 3 1 1
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/x6pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; cs="$1"; shift
  timeout 200 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $O/$name -o p -- python3 $R/tools/x6/one.py "$@" > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py $O/$name.json $O/$name/p_counter_collection.csv && rm -rf $O/$name; }
run sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "$@"
run sq2 "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "$@"
python3 - <<PY
import json
for n in ("sq","sq2"):
    d=json.load(open("$O/%s.json"%n))
    for k,v in d.items():
        if "x6" in k:
            print(n,k,{c:round(x["avg"],1) for c,x in v.items()})
PY
