#!/bin/bash
# One-file variant of the library: recompiles ONLY the named source with extra defines and links it with the shipped objects of
# every other file: tools/build_variant.sh TAG ffq_wlinear.hip "-DFFQ_WL_CLUSTER_MODE=0" -> tools/_exp/libffq_TAG.so (git-ignored).
# Use with FFQ_LIB=tools/_exp/libffq_TAG.so python tools/{wq_time,gemm_time,...}.py  (A/B of kernel variants in one gpurun call).
set -e
TAG=$1; SRC=$2; EXTRA=${3:-}
cd "$(dirname "$0")/../fastforward_amd/csrc"
make -s >/dev/null
OUT=../../tools/_exp; mkdir -p $OUT/$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function $EXTRA"
/opt/rocm/bin/hipcc $FLAGS -c $SRC -o $OUT/$TAG/${SRC%.hip}.o
OBJS=""
for f in _build/*.o; do
  if [ "$(basename $f)" = "${SRC%.hip}.o" ]; then OBJS="$OBJS $OUT/$TAG/${SRC%.hip}.o"; else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libffq_$TAG.so $OBJS
rm -rf $OUT/$TAG
ls -la $OUT/libffq_$TAG.so
