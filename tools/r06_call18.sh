#!/bin/bash
# round 6, call 18: the gate + up + SiLU*up launch of the weight-only path on the one-wave-per-SIMD kernel (shipped) against the 8-wave
# kernel (no4wmlp): bit-equality first, then timings at 16 k / 4 k tokens, interleaved rounds
mkdir -p gpurun_out/r06
{
timeout 600 python tools/w4_mlp_check.py 2>&1 | grep -v amdgpu
for round in 1 2; do
  for tag in shipped no4wmlp; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib timeout 300 python tools/mlp_wq_time.py 16384 2>&1 | grep -v amdgpu
    env $lib timeout 300 python tools/mlp_wq_time.py 4096 2>&1 | grep -v amdgpu
  done
done
} > gpurun_out/r06/w4_mlp_ab.txt 2>&1
cat gpurun_out/r06/w4_mlp_ab.txt
