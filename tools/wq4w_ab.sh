#!/bin/bash
# A/B of library variants of the weight-only GEMM in one gpurun call, interleaved rounds: tools/wq4w_ab.sh "T ..." tag[:ENV=..] ...
# (tag "default" = the shipped library; any other tag = tools/_exp/libffq_<tag>.so from tools/build_variant.sh)
TS=$1; shift
for round in 1 2; do
  for v in "$@"; do
    tag=${v%%:*}; envs=""; [ "$v" != "$tag" ] && envs=${v#*:}
    lib=""; [ "$tag" != default ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    for T in $TS; do
      echo "== $v T=$T round $round"
      env $lib $envs timeout 300 python tools/wq_time.py $T 2>&1 | grep -v amdgpu | sed -E 's/w8 one-pass [^|]*\|//; s/w4g128 one-pass [^|]*\|//; s/w4g128 packed [^|]*\|//; s/mlp_gate_up_wq one-pass [^|]*\|//' | cut -c1-400
    done
  done
done
