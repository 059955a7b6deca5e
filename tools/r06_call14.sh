#!/bin/bash
# round 6, call 14: GPU suite on the tree with the per-XCD contraction start (power-of-two depths >= 8192), then the int8 GEMM on both models' shapes
mkdir -p gpurun_out/r06
( timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > gpurun_out/r06/gputests_call14.txt
tail -4 gpurun_out/r06/gputests_call14.txt
{
echo "=== 8B shapes, 16384 tokens"; timeout 300 python tools/gemm_time.py 16384 2>&1 | grep -v amdgpu
echo "=== 70B shapes, 8192 tokens"; GT_MODEL=70b timeout 300 python tools/gemm_time.py 8192 2>&1 | grep -v amdgpu
echo "=== bf16-image GEMM"; KS="4096 8192 14336 16384" timeout 600 python tools/wq_k_sweep.py 16384 2>&1 | grep -v amdgpu
} > gpurun_out/r06/krot_shipped.txt 2>&1
cat gpurun_out/r06/krot_shipped.txt
