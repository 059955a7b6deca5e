#!/bin/bash
# Whole-library experiment builds: tools/lib_variants.sh TAG "-DFLAG ..." [...] -> fastforward_amd/csrc/_build/libffq_TAG.so
set -e
cd "$(dirname "$0")/../fastforward_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
while [ $# -ge 2 ]; do
  TAG=$1; DEFS=$2; shift 2
  mkdir -p _build/$TAG
  for f in ffq_*.hip; do ( /opt/rocm/bin/hipcc $FLAGS $DEFS -c $f -o _build/$TAG/${f%.hip}.o ) & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _build/libffq_$TAG.so _build/$TAG/*.o && rm -rf _build/$TAG && echo built $TAG
done
