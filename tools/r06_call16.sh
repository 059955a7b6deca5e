#!/bin/bash
# round 6, call 16: GPU suite on the tree with the half-step-ahead schedule as the shipped form; weight-only timings at 16 k / 4 k tokens; cfg2 / cfg4
mkdir -p gpurun_out/r06
( timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > gpurun_out/r06/gputests_call16.txt
tail -4 gpurun_out/r06/gputests_call16.txt
: > gpurun_out/r06/wq_time_call16.txt
for T in 16384 4096; do timeout 300 python3 tools/wq_time.py $T 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/wq_time_call16.txt; done
cut -c1-400 gpurun_out/r06/wq_time_call16.txt
timeout 2400 python3 tools/bench_configs.py --out gpurun_out/r06/configs_call16.json > gpurun_out/r06/configs_call16.log 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06/configs_call16.json'))
for c in ('cfg2','cfg4'):
    for k,v in d[c].items():
        if isinstance(v,dict) and 'tokens_per_s' in v: print(c,k,v['ms'],v['tokens_per_s'])
PY
