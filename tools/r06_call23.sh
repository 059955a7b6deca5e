#!/bin/bash
# round 6, call 23: residual add + RMSNorm + quantize with blocks striding over the rows and the next row's loads in flight (rmspN = N blocks
# per CU) against one block per row (shipped)
mkdir -p gpurun_out/r06
{
for round in 1 2; do
  for tag in shipped rmsp4 rmsp8 rmsp16; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib timeout 300 python tools/rms_time.py 2>&1 | grep -v amdgpu
  done
done
} > gpurun_out/r06/rms_persist_ab.txt 2>&1
cat gpurun_out/r06/rms_persist_ab.txt
