#!/bin/bash
# round 6, call 24: the two-pass plan rule (from 1536 tokens where the launch has >= 144 tiles): GPU suite, the library's own choice beside both
# forced forms at 1536 / 2048 / 3072 tokens, and cfg2 / cfg4 at B = 1
mkdir -p gpurun_out/r06
( timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > gpurun_out/r06/gputests_call24.txt
tail -3 gpurun_out/r06/gputests_call24.txt
timeout 600 python - > gpurun_out/r06/twopass_rule.txt 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, '.')
from fastforward_amd import ops
from bench import event_time_ms
t = lambda fn: event_time_ms(lambda r: fn(r), iters=6, reps=6) * 1e3
dev = "cuda"
for T in (1024, 1536, 2048, 3072):
    x = torch.randn(T, 4096, device=dev, dtype=torch.bfloat16)
    ws = [(torch.randint(-128, 128, (n, 4096), device=dev, dtype=torch.int8), torch.rand(n, device=dev) * 1e-3 + 1e-4) for n in (4096, 1024, 1024)]
    row = []
    for tp in (None, False, True):
        row.append(t(lambda r: ops.linear_wq_multi(x, [w for w, _ in ws], [s for _, s in ws], [None] * 3, two_pass=tp)))
    print(f"T={T:5d} q/k/v plan {row[0]:7.1f}us | one-pass {row[1]:7.1f} | two-pass {row[2]:7.1f}")
    w, s = ws[0]
    row = [t(lambda r: ops.linear_wq(x, w, s, None, two_pass=tp)) for tp in (None, False, True)]
    print(f"T={T:5d} o     plan {row[0]:7.1f}us | one-pass {row[1]:7.1f} | two-pass {row[2]:7.1f}")
PY
cat gpurun_out/r06/twopass_rule.txt | grep -v amdgpu
timeout 1500 python3 tools/bench_configs.py --out gpurun_out/r06/configs_call24.json > gpurun_out/r06/configs_call24.log 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06/configs_call24.json'))
for c in ('cfg2','cfg4'):
    for k,v in d[c].items():
        if isinstance(v,dict) and 'tokens_per_s' in v: print(c,k,v['ms'],v['tokens_per_s'])
PY
