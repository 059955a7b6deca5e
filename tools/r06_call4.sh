#!/bin/bash
# round 6, call 4: GPU suite with the LDS-DMA form of the 128-column tiles; sweeps: shipped (ring 6) / ring 4 / ring 3 / register-staged form
mkdir -p gpurun_out/r06
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r06/gputests_call4.txt
{
for tag in shipped ring4 ring3 nodma; do
  lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
  echo "=== $tag: few rows"
  env $lib timeout 900 python tools/wq_skinny_sweep.py 32 64 128 2>&1 | grep -v amdgpu
  echo "=== $tag: 129 .. 1024 rows"
  env $lib timeout 900 python tools/wq_split_sweep.py 256 512 2>&1 | grep -v amdgpu
done
} > gpurun_out/r06/wq_mid_sweep_v3.txt 2>&1
tail -12 gpurun_out/r06/gputests_call4.txt
