#!/bin/bash
# A/B of int8 GEMM library variants in one gpurun call, interleaved rounds: tools/gemm_ab.sh tag ... ("default" = shipped library)
for round in 1 2 3; do
  for tag in "$@"; do
    lib=""; [ "$tag" != default ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "== $tag round $round"
    env $lib timeout 300 python tools/gemm_time.py 16384 2>&1 | grep -v amdgpu
  done
done
