#!/bin/bash
# A/B of weight-only GEMM builds on one box: tools/wq_ab.sh TAG...
cd "$(dirname "$0")/.."
for round in 1 2; do
  echo "== base (round $round)"; python3 tools/wq_time.py 16384 2>&1 | grep -v amdgpu.ids | tail -1
  for tag in "$@"; do
    echo "== $tag (round $round)"; FFQ_LIB=fastforward_amd/csrc/_build/libffq_$tag.so python3 tools/wq_time.py 16384 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
