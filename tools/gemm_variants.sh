#!/bin/bash
# Build experiment variants of the GEMM translation unit: tools/gemm_variants.sh TAG "-DFLAG ..." [TAG2 "..."]...
# -> fastforward_amd/csrc/_build/libffq_TAG.so (other objects reused). Time with FFQ_LIB=<that> tools/gemm_time.py
set -e
cd "$(dirname "$0")/../fastforward_amd/csrc"
make -s -j8
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function"
while [ $# -ge 2 ]; do
  TAG=$1; DEFS=$2; shift 2
  ( /opt/rocm/bin/hipcc $FLAGS $DEFS -c ffq_linear.hip -o _build/ffq_linear_$TAG.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _build/libffq_$TAG.so $(ls _build/ffq_*.o | grep -v "ffq_linear\.o\|ffq_linear_") _build/ffq_linear_$TAG.o && echo built $TAG ) &
done
wait
