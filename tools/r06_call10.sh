#!/bin/bash
# round 6, call 10: 128-column tiles with 8 waves of 16 columns (shipped) against 4 waves of 32 (waves4); GPU suite first
mkdir -p gpurun_out/r06
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r06/gputests_call10.txt
tail -4 gpurun_out/r06/gputests_call10.txt
{
for round in 1 2; do
  for tag in shipped waves4; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag round $round"
    env $lib timeout 900 python tools/wq_rows_sweep.py 33 64 128 256 512 2>&1 | grep -v amdgpu
  done
done
} > gpurun_out/r06/wq_mid_waves_ab.txt 2>&1
cut -c1-170 gpurun_out/r06/wq_mid_waves_ab.txt
