#!/bin/bash
# round 6, call 3: the whole GPU suite (all failures listed), the 128-column tiles with straight-line waits (v2) + a deeper code ring (md62), A3 variants
mkdir -p gpurun_out/r06
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > gpurun_out/r06/gputests_call3.txt
{
for tag in shipped md62; do
  lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
  echo "=== $tag: few rows"
  env $lib timeout 900 python tools/wq_skinny_sweep.py 24 32 64 128 2>&1 | grep -v amdgpu
  echo "=== $tag: 129 .. 1024 rows"
  env $lib timeout 900 python tools/wq_split_sweep.py 129 256 512 1024 2>&1 | grep -v amdgpu
done
} > gpurun_out/r06/wq_mid_sweep_v2.txt 2>&1
{
for round in 1 2; do
  for tag in shipped dynw1 dynw2 dynb4; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag (round $round)"
    env $lib timeout 300 python tools/a3_time.py 2>&1 | grep -v amdgpu | grep -i "per-token\|per-channel"
  done
done
} > gpurun_out/r06/a3_variants.txt 2>&1
cat gpurun_out/r06/gputests_call3.txt | tail -25
