#!/bin/bash
# round 6, call 2: the GPU suite with the 128-column tiles in, sweeps of the new form against the skinny form (variant sk129: FFQ_MID_MIN_M=129), the 256-row tiles and the vendor GEMM
mkdir -p gpurun_out/r06
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06/gputests_call2.txt
{
echo "=== shipped: few rows"
timeout 900 python tools/wq_skinny_sweep.py 8 16 24 32 64 128 2>&1 | grep -v amdgpu
echo "=== sk129 (skinny form up to 128 rows, round 5's choice): few rows"
FFQ_LIB=tools/_exp/libffq_sk129.so timeout 900 python tools/wq_skinny_sweep.py 16 24 32 64 128 2>&1 | grep -v amdgpu
echo "=== shipped: 129 .. 1024 rows"
timeout 900 python tools/wq_split_sweep.py 129 256 384 512 1024 2>&1 | grep -v amdgpu
} > gpurun_out/r06/wq_mid_sweep.txt 2>&1
cat gpurun_out/r06/gputests_call2.txt
