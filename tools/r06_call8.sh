#!/bin/bash
# round 6, call 8: GPU suite; the one-wave-per-SIMD bf16 GEMM with its A and B LDS-DMA pieces four MFMAs apart (w4spread) against the shipped placement, interleaved rounds; full bench line
mkdir -p gpurun_out/r06
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r06/gputests_call8.txt
tail -4 gpurun_out/r06/gputests_call8.txt
{
for round in 1 2 3; do
  for tag in shipped w4spread; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "== $tag round $round"
    env $lib timeout 300 python tools/wq_time.py 16384 2>&1 | grep -v amdgpu | cut -c1-420
  done
done
} > gpurun_out/r06/w4_spread_ab.txt 2>&1
grep "layer mix\|^==" gpurun_out/r06/w4_spread_ab.txt | cut -c1-200
timeout 900 python bench.py > gpurun_out/r06/bench_call8.json 2> gpurun_out/r06/bench_call8.err
python -c "
import json
d=json.loads(open('gpurun_out/r06/bench_call8.json').read().strip().splitlines()[-1]); print(d['summary']); print(d['host_us_per_op']); print(d['roofline']['per_shape'])"
