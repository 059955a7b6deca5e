#!/bin/bash
# round 6, call 17: int8 GEMM with two LDS-DMA pieces in every load segment (i8spread) against the shipped phases-0-and-1 schedule:
# bit-exactness first (8B shapes; ragged + small K + 70B through the test file's own shapes), then timings, interleaved rounds
mkdir -p gpurun_out/r06
{
FFQ_LIB=tools/_exp/libffq_i8spread.so timeout 600 python tools/gemm_check.py 2>&1 | grep -v amdgpu
for round in 1 2 3; do
  for tag in shipped i8spread; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib timeout 300 python tools/gemm_time.py 16384 2>&1 | grep -v amdgpu
  done
done
for tag in shipped i8spread; do
  lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
  echo "=== $tag (70B shapes, 8192 tokens)"
  env $lib GT_MODEL=70b timeout 300 python tools/gemm_time.py 8192 2>&1 | grep -v amdgpu
done
} > gpurun_out/r06/i8_spread_ab.txt 2>&1
cat gpurun_out/r06/i8_spread_ab.txt
