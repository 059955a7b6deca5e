#!/bin/bash
# round 6, call 9: does an L2 warm-up 2 / 4 super-steps ahead of the LDS-DMA help the 14336-deep contraction (down_proj) where it cost 8 % on the 4096-deep ones (round 5)?
mkdir -p gpurun_out/r06
{
for round in 1 2; do
  for tag in shipped w4ahead2 w4ahead4; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    for T in 16384 4096; do
      echo "== $tag round $round T=$T"
      env $lib timeout 300 python tools/wq_time.py $T 2>&1 | grep -v amdgpu | grep "^qo\|^down\|^gateup" | awk -F'|' '{print $1 "|" $2 "|" $6 "|" $7}' | cut -c1-220
    done
  done
done
} > gpurun_out/r06/w4_ahead_ab.txt 2>&1
cat gpurun_out/r06/w4_ahead_ab.txt
