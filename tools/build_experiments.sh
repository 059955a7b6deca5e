#!/bin/bash
# Tuning build of the library with -DFFQ_EXPERIMENTS (environment-controlled launch parameters that the shipped library does
# not have): tools/_exp/libffq_exp.so. Use with FFQ_LIB=tools/_exp/libffq_exp.so python tools/wq_time.py ...
set -e
cd "$(dirname "$0")/../fastforward_amd/csrc"
mkdir -p ../../tools/_exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function -DFFQ_EXPERIMENTS $EXTRA"
for f in ffq_wlinear ffq_linear; do /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o ../../tools/_exp/$f.o & done
wait
OBJS=$(ls _build/*.o | grep -v "ffq_wlinear.o\|ffq_linear.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_exp/libffq_exp.so $OBJS ../../tools/_exp/ffq_wlinear.o ../../tools/_exp/ffq_linear.o
ls -la ../../tools/_exp/libffq_exp.so
