#!/bin/bash
# A/B of the one-wave-per-SIMD kernel against the 8-wave one and of its tile-walk parameters, one gpurun call, interleaved rounds
for round in 1 2; do
  for v in "default:" "no4w:FFQ_LIB=tools/_exp/libffq_no4w.so" "gm8:FFQ_LIB=tools/_exp/libffq_w4x.so FFQ_WQ_GROUP_M=8" "gm2:FFQ_LIB=tools/_exp/libffq_w4x.so FFQ_WQ_GROUP_M=2" "cols8:FFQ_LIB=tools/_exp/libffq_w4x.so FFQ_WQ_GROUP_COLS=1 FFQ_WQ_GROUP_M=8"; do
    tag=${v%%:*}; envs=${v#*:}
    for T in 16384 4096; do
      echo "== $tag T=$T round $round"
      env $envs timeout 300 python tools/wq_time.py $T 2>&1 | grep -v amdgpu | sed -E 's/w8 one-pass [^|]*\|//; s/w4g128 one-pass [^|]*\|//; s/w4g128 packed [^|]*\|//; s/mlp_gate_up_wq one-pass [^|]*\|//' | cut -c1-400
    done
  done
done
