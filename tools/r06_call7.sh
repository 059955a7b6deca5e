#!/bin/bash
# round 6, call 7: GPU suite (q/k/v as one int8 launch in), bench line old (three launches) vs new (one launch) on ONE box, rows sweep of the final plan
mkdir -p gpurun_out/r06
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r06/gputests_call7.txt
tail -5 gpurun_out/r06/gputests_call7.txt
timeout 900 python bench.py --no-side-measurements > gpurun_out/r06/bench_call7_qkv1.json 2> gpurun_out/r06/bench_call7.err
timeout 900 python bench.py --no-side-measurements --qkv-three-launches > gpurun_out/r06/bench_call7_qkv3.json 2>> gpurun_out/r06/bench_call7.err
timeout 900 python bench.py --no-side-measurements > gpurun_out/r06/bench_call7_qkv1_b.json 2>> gpurun_out/r06/bench_call7.err
timeout 900 python bench.py --no-side-measurements --qkv-three-launches > gpurun_out/r06/bench_call7_qkv3_b.json 2>> gpurun_out/r06/bench_call7.err
for f in qkv1 qkv3 qkv1_b qkv3_b; do python -c "
import json,sys
d=json.loads(open('gpurun_out/r06/bench_call7_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['calibration']['sequences_per_s_all_gpus'])"; done
timeout 1200 python tools/wq_rows_sweep.py 17 24 32 2>&1 | grep -v amdgpu > gpurun_out/r06/wq_rows_sweep_small.txt
cat gpurun_out/r06/wq_rows_sweep_small.txt | cut -c1-200
