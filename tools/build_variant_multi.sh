#!/bin/bash
# Variant of the library with SEVERAL sources recompiled under extra defines, linked with the shipped objects of the rest:
#   tools/build_variant_multi.sh TAG "-DFFQ_TICKET_ORDER=__ATOMIC_RELAXED" ffq_wskinny.hip ffq_wlinear.hip ...  -> tools/_exp/libffq_TAG.so
# (one-file form: tools/build_variant.sh). Use with FFQ_LIB=tools/_exp/libffq_TAG.so in one gpurun call beside the shipped library.
set -e
TAG=$1; EXTRA=$2; shift 2
cd "$(dirname "$0")/../fastforward_amd/csrc"
make -s >/dev/null
OUT=../../tools/_exp; mkdir -p $OUT/$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function $EXTRA"
for SRC in "$@"; do /opt/rocm/bin/hipcc $FLAGS -c $SRC -o $OUT/$TAG/${SRC%.hip}.o & done
wait
OBJS=""
for f in _build/*.o; do
  b=$(basename $f)
  if [ -f $OUT/$TAG/$b ]; then OBJS="$OBJS $OUT/$TAG/$b"; else OBJS="$OBJS $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libffq_$TAG.so $OBJS
rm -rf $OUT/$TAG
ls -la $OUT/libffq_$TAG.so
