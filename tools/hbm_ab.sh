#!/bin/bash
# A/B of library variants on the HBM-bound kernels: tools/hbm_ab.sh "filter" TAG [TAG ...]   (two interleaved rounds)
F=$1; shift
for round in 1 2; do
  for tag in "$@"; do
    echo "== $tag round $round"
    FFQ_LIB=tools/_exp/libffq_$tag.so timeout 300 python tools/hbm_time.py "$F" 2>&1 | grep -v amdgpu.ids
  done
done
