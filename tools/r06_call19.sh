#!/bin/bash
# round 6, call 19: GPU suite with the MLP mode of the one-wave-per-SIMD kernel shipped; cfg2 / cfg4 forwards against their vendor arms
mkdir -p gpurun_out/r06
( timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > gpurun_out/r06/gputests_call19.txt
tail -4 gpurun_out/r06/gputests_call19.txt
timeout 2400 python3 tools/bench_configs.py --out gpurun_out/r06/configs_call19.json > gpurun_out/r06/configs_call19.log 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06/configs_call19.json'))
for c in ('cfg2','cfg4'):
    for k,v in d[c].items():
        if isinstance(v,dict) and 'tokens_per_s' in v: print(c,k,v['ms'],v['tokens_per_s'])
PY
