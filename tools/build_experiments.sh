#!/bin/bash
# Tuning build of the library with -DFFQ_EXPERIMENTS (environment-controlled launch parameters that the shipped library does
# not have) plus any extra defines: tools/build_experiments.sh [TAG] ["-DFLAG ..."] -> tools/_exp/libffq_TAG.so (git-ignored).
# Use with FFQ_LIB=tools/_exp/libffq_TAG.so python tools/{wq_time,gemm_time,mlp_wq_time,attn_time,...}.py
set -e
TAG=${1:-exp}; EXTRA=${2:-}
cd "$(dirname "$0")/../fastforward_amd/csrc"
OUT=../../tools/_exp; mkdir -p $OUT/$TAG
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function -I../../include -DFFQ_EXPERIMENTS $EXTRA"
for f in *.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o $OUT/$TAG/${f%.hip}.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libffq_$TAG.so $OUT/$TAG/*.o
rm -rf $OUT/$TAG
ls -la $OUT/libffq_$TAG.so
