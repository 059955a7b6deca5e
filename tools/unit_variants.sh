#!/bin/bash
# Experiment builds of ONE translation unit: tools/unit_variants.sh ffq_producers TAG "-DFLAG ..." [TAG2 "..."]...
# -> fastforward_amd/csrc/_build/libffq_TAG.so (the other objects are the shipped ones). A/B with tools/arith_ab.py <that>.
set -e
cd "$(dirname "$0")/../fastforward_amd/csrc"
make -s -j8
UNIT=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
while [ $# -ge 2 ]; do
  TAG=$1; DEFS=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $DEFS -c $UNIT.hip -o _build/${UNIT}_$TAG.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _build/libffq_$TAG.so $(ls _build/ffq_*.o | grep -v "${UNIT}\.o\|_build/ffq_[a-z0-9]*_") _build/${UNIT}_$TAG.o && echo built $TAG ) &
done
wait
