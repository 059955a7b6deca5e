#!/bin/bash
# round 6, call 5: GPU suite on the split ops package; 128-column tiles: 64-row tiles at every row count with 3 / 2 blocks per CU against ring 3 (BM 128, 2 blocks per CU)
mkdir -p gpurun_out/r06
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r06/gputests_call5.txt
{
for tag in bm64r3 bm64r4 ring3; do
  lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
  echo "=== $tag: few rows"
  env $lib timeout 900 python tools/wq_skinny_sweep.py 128 2>&1 | grep -v amdgpu
  echo "=== $tag: 129 .. 1024 rows"
  env $lib timeout 900 python tools/wq_split_sweep.py 256 512 2>&1 | grep -v amdgpu
done
} > gpurun_out/r06/wq_mid_sweep_v4.txt 2>&1
tail -6 gpurun_out/r06/gputests_call5.txt
