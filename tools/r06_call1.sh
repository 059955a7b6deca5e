#!/bin/bash
# round 6, call 1: the GPU suite on the tree, the ticket-order A/B (ACQ_REL = shipped, relaxed = round 5) on one box, the bench line with the same-box probe
mkdir -p gpurun_out/r06
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r06/gputests_call1.txt
{
for round in 1 2; do
  for tag in shipped relaxed; do
    lib=""; [ "$tag" != shipped ] && lib="FFQ_LIB=tools/_exp/libffq_$tag.so"
    echo "=== $tag (round $round): skinny sweep"
    env $lib timeout 600 python tools/wq_skinny_sweep.py 1 64 128 2>&1 | grep -v amdgpu | cut -c1-200
    echo "=== $tag (round $round): split sweep T=512"
    env $lib timeout 600 python tools/wq_split_sweep.py 512 2>&1 | grep -v amdgpu
    echo "=== $tag (round $round): min/max"
    env $lib timeout 300 python tools/minmax_time.py 2>&1 | grep -v amdgpu
  done
done
} > gpurun_out/r06/ticket_order_ab.txt 2>&1
timeout 900 python bench.py > gpurun_out/r06/bench_call1.json 2> gpurun_out/r06/bench_call1.err
tail -c 1500 gpurun_out/r06/bench_call1.json
cat gpurun_out/r06/gputests_call1.txt
