#!/bin/bash
# round 6, call 15: the bf16-image GEMM with the A operand half a step ahead of B (LDS-DMA spread over both k-halves) against the shipped
# slot-at-a-time schedule: bit-equality with the 8-wave kernel first, then the K sweep and the layer shapes, interleaved rounds
mkdir -p gpurun_out/r06
{
FFQ_LIB=tools/_exp/libffq_w4half.so timeout 600 python tools/w4_check.py 2>&1 | grep -v amdgpu
for round in 1 2; do
  for tag in shipped w4half; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib KS="2048 4096 8192 12288 14336 16384" timeout 600 python tools/wq_k_sweep.py 16384 2>&1 | grep -v amdgpu
    env $lib KS="4096 14336" timeout 600 python tools/wq_k_sweep.py 4096 2>&1 | grep -v amdgpu
  done
done
} > gpurun_out/r06/w4_half_ab.txt 2>&1
cat gpurun_out/r06/w4_half_ab.txt
