#!/bin/bash
# A/B of GEMM experiment builds on one box: tools/gemm_ab.sh TAG... (built by tools/gemm_variants.sh) -> gpurun_out/gemm_ab.log
# Each build is timed twice, interleaved with the shipped library ("base"), on the four Llama-3-8B shapes at T = 16384.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2; do
  echo "== base (round $round)"; python3 tools/gemm_time.py 16384 2>&1 | grep -v amdgpu.ids
  for tag in "$@"; do
    echo "== $tag (round $round)"; FFQ_LIB=fastforward_amd/csrc/_build/libffq_$tag.so python3 tools/gemm_time.py 16384 2>&1 | grep -v amdgpu.ids
  done
done
