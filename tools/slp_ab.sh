#!/bin/bash
# A/B on one box: every non-convolution source built with / without the SLP vectoriser (mano.hip is always built without it).
# usage (GPU box): bash tools/slp_ab.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/slp_ab; mkdir -p $O; cd $R/dsf_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
build() {  # $1 = extra flag for the non-conv files
  for f in raster pfd hand_geom image_ops data_ops norm loss optim pool volume; do /opt/rocm/bin/hipcc $FLAGS $1 -c $f.hip -o ../lib/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdsf_hip.so ../lib/*.o
}
: > $O/ab.txt
for round in 1 2; do
  for v in slp noslp; do
    if [ $v = slp ]; then build ""; else build "-fno-slp-vectorize"; fi
    for c in 2 5; do
      (cd $R; timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v config $c', d['value'], d['ms_per_step'])" >> $O/ab.txt)
    done
  done
done
cat $O/ab.txt
