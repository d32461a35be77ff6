#!/bin/bash
# Builds libdsf_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# -ffp-contract=off: coverage / argmin decisions must be bit-identical to the oracle's
# separate IEEE ops; fused multiply-adds are written explicitly (fmaf) where wanted.
# A failed compile fails the build: the stale object is removed before compiling, every background
# job's status is collected, and a change of FLAGS rebuilds everything.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT
# -fno-slp-vectorize, for the whole library.  Found with mano.hip in round 4: its SLP-vectorised skinning loops (v_pk_fma_f32 on
# ds_read_b128 broadcasts) returned wrong bits in lanes 48-63 whenever conv_x6 workgroups shared the CU, the scalar form never did
# (profiles/r04_mano_beside_conv_x6.txt, tools/platform/mano_beside_conv_x6.py,
# tests/test_gpu_determinism.py::test_mano_backward_is_stable_beside_convolution_workgroups).  The trigger was not isolated beyond
# "SLP-vectorised code beside conv_x6", and every kernel of a step runs beside the conv_x6 workgroups of the weight-gradient stream,
# so every kernel gets the form that never failed.  It costs nothing: the non-convolution sources +0.03 ms of a 19.5 ms step, the
# convolution sources -0.1 ... -0.2 ms (same box, alternating: 20.22 / 20.24 / 20.26 ms with, 20.15 / 20.12 / 20.07 without).
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function"
SRCS="api mano raster pfd hand_geom image_ops data_ops conv conv_x6 conv_c1 norm loss optim pool volume"
if [ "$(cat $OUT/.flags 2>/dev/null)" != "$FLAGS" ]; then
  rm -f $OUT/*.o
  echo "$FLAGS" > $OUT/.flags
fi
pids=()
for f in $SRCS; do
  if [ ! -f $OUT/$f.o ] || [ $f.hip -nt $OUT/$f.o ] || [ common.h -nt $OUT/$f.o ] || [ ../../include/dsf_hip.h -nt $OUT/$f.o ]; then
    rm -f $OUT/$f.o
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/$f.o &
    pids+=($!)
  fi
done
fail=0
for p in "${pids[@]}"; do
  wait $p || fail=1
done
if [ $fail -ne 0 ]; then
  echo "build.sh: a compile failed" >&2
  exit 1
fi
OBJS=""
for f in $SRCS; do OBJS="$OBJS $OUT/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdsf_hip.so $OBJS
echo built $OUT/libdsf_hip.so
