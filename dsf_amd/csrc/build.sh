#!/bin/bash
# Builds libdsf_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# -ffp-contract=off: coverage / argmin decisions must be bit-identical to the oracle's
# separate IEEE ops; fused multiply-adds are written explicitly (fmaf) where wanted.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for f in api mano raster pfd hand_geom image_ops data_ops conv conv_x6 conv_c1 norm loss optim; do
  if [ ! -f $OUT/$f.o ] || [ $f.hip -nt $OUT/$f.o ] || [ common.h -nt $OUT/$f.o ] || [ ../../include/dsf_hip.h -nt $OUT/$f.o ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o $OUT/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdsf_hip.so $OUT/api.o $OUT/mano.o $OUT/raster.o $OUT/pfd.o $OUT/hand_geom.o $OUT/image_ops.o $OUT/data_ops.o $OUT/conv.o $OUT/conv_x6.o $OUT/conv_c1.o $OUT/norm.o $OUT/loss.o $OUT/optim.o
echo built $OUT/libdsf_hip.so
