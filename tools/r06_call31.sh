#!/bin/bash
# round 6, call 31: the 128-column tiles with a ring twice as deep where the launch has at most one block per CU (shipped) against the
# 3 / 4-stage ring everywhere (nodeep): mid-GPU tests first, then the rows sweep at 17 .. 256 rows, interleaved rounds
mkdir -p gpurun_out/r06
( timeout 900 python -m pytest tests/test_mid_gpu.py tests/test_skinny_gpu.py -m gpu -q -x 2>&1 | tail -3 ) > gpurun_out/r06/gputests_call31.txt
tail -2 gpurun_out/r06/gputests_call31.txt
{
for round in 1 2; do
  for tag in shipped nodeep; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib timeout 900 python tools/wq_rows_sweep.py 17 33 64 128 256 2>&1 | grep -v amdgpu | cut -c1-75
  done
done
} > gpurun_out/r06/wq_mid_deep_ab.txt 2>&1
cat gpurun_out/r06/wq_mid_deep_ab.txt
