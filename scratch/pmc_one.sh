#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a -o p -- python3 $R/scratch/one_conv.py > $O/a.log 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/a/p_counter_collection.csv")))
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'igemm' in r['Kernel_Name']:
        key = (r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size'])
        d[key][r['Counter_Name']].append(float(r['Counter_Value']))
        d[key]['dur'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, c in d.items():
    n = len(c['GRBM_GUI_ACTIVE']); av = lambda x: sum(c[x]) / max(len(c[x]), 1)
    dur = sum(c['dur']) / len(c['dur']); cyc = av('GRBM_GUI_ACTIVE') / 8
    print(k, 'n', n, f"dur {dur/1e3:.1f}us clk {cyc/dur:.3f} GHz  mfma_util {av('SQ_VALU_MFMA_BUSY_CYCLES')/(cyc*1024):.3f}  waves/simd {av('SQ_WAVE_CYCLES')*4/(cyc*1024):.2f} wait_any {av('SQ_WAIT_ANY')/av('SQ_WAVE_CYCLES'):.2f} wait_inst {av('SQ_WAIT_INST_ANY')/av('SQ_WAVE_CYCLES'):.2f} active {av('SQ_ACTIVE_INST_ANY')/av('SQ_WAVE_CYCLES'):.2f}")
PY
tail -6 $O/a.log; rm -rf $O/a
