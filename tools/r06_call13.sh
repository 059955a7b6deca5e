#!/bin/bash
# round 6, call 13: per-XCD contraction start on the depths whose row pitch is a large power of two (70B: K = 8192, 28672; 16384), bf16-image
# GEMM and int8 GEMM; krot1 = XCD x starts at x/8 of the depth, krot3 = the same plus 5 x super-steps (not a multiple of 4 KiB apart)
mkdir -p gpurun_out/r06
{
for round in 1 2; do
  for tag in shipped w4krot1 w4krot3; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib KS="8192 16384 28672 14336" timeout 600 python tools/wq_k_sweep.py 16384 2>&1 | grep -v amdgpu
  done
done
for round in 1 2; do
  for tag in shipped i8krot1 i8krot3; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round (70B shapes, 8192 tokens)"
    env $lib GT_MODEL=70b timeout 300 python tools/gemm_time.py 8192 2>&1 | grep -v amdgpu
  done
done
} > gpurun_out/r06/krot_ab2.txt 2>&1
cat gpurun_out/r06/krot_ab2.txt
