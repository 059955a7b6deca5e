#!/bin/bash
# A/B of library variants (tools/build_variant.sh) in ONE gpurun call, interleaved rounds: tools/wq_ab.sh "T..." TAG [TAG ...]
TS=$1; shift
for round in 1 2; do
  for tag in "$@"; do
    for T in $TS; do
      echo "== $tag T=$T round $round"
      FFQ_LIB=tools/_exp/libffq_$tag.so timeout 300 python tools/wq_time.py $T 2>&1 | grep -E "layer mix|gate\+up" | cut -c1-700
    done
  done
done
