#!/bin/bash
# round 6, call 12: every XCD (krot1) / every block (krot2) walks the contraction from its own starting depth — does the stride-dependent
# part of the gap to the vendor's GEMM come from all CUs requesting the same depth of every row at the same time?
mkdir -p gpurun_out/r06
{
for round in 1 2; do
  for tag in shipped w4krot1 w4krot2; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib KS="4096 8192 12288 14336 16384" timeout 600 python tools/wq_k_sweep.py 16384 2>&1 | grep -v amdgpu
  done
done
for round in 1 2; do
  for tag in shipped i8krot1 i8krot2; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib timeout 300 python tools/gemm_time.py 16384 2>&1 | grep -v amdgpu
  done
done
} > gpurun_out/r06/krot_ab.txt 2>&1
cat gpurun_out/r06/krot_ab.txt
