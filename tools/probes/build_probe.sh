#!/bin/bash
# Builds a MEASUREMENT variant of libdsf_hip.so: the timing probes (tools/probes/*.patch: -DDSF_WRW_PROBE=n,
# -DDSF_X6_APPLY_PROBE, -DDSF_X6_TAPS_OUTER, -DDSF_BN_MAX_WGS=n) are applied to a scratch copy of dsf_amd/csrc, compiled with
# the given flags and the result replaces dsf_amd/lib/libdsf_hip.so until `build_probe.sh --restore` (or csrc/build.sh after a
# source change) puts the product library back.  The product sources and the product build carry no probe code or flag hook.
#   usage: tools/probes/build_probe.sh -DDSF_WRW_PROBE=2        tools/probes/build_probe.sh --restore
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
LIB=$R/dsf_amd/lib
if [ "$1" = "--restore" ]; then
  rm -f $LIB/libdsf_hip.so; bash $R/dsf_amd/csrc/build.sh; exit 0
fi
S=$(mktemp -d /tmp/dsf_probe.XXXXXX)
mkdir -p $S/dsf_amd/lib $S/include
cp -r $R/dsf_amd/csrc $S/dsf_amd/csrc
cp $R/include/*.h $S/include/
( cd $S && for p in $R/tools/probes/*.patch; do patch -s -p1 < $p; done )
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $*"
cd $S/dsf_amd/csrc
OBJS=""
for f in *.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o $S/dsf_amd/lib/${f%.hip}.o & OBJS="$OBJS $S/dsf_amd/lib/${f%.hip}.o"; done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB/libdsf_hip.so $OBJS
rm -rf $S
echo "probe library in place ($*): results may be numerically wrong; restore with tools/probes/build_probe.sh --restore"
