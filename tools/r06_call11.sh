#!/bin/bash
# round 6, call 11: the backward op's one-launch route for small problems (parity + host microseconds), and the bf16-image GEMM of the
# weight-only path as a function of the contraction depth next to the vendor's (what the down_proj gap depends on)
mkdir -p gpurun_out/r06
( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_seam1_ops.py tests/test_torch_extension.py -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r06/gputests_call11.txt
tail -3 gpurun_out/r06/gputests_call11.txt
timeout 300 python - > gpurun_out/r06/host_us_call11.txt 2>&1 <<'PY'
import json, torch, bench
print(json.dumps(bench.host_us_per_op(torch.device("cuda:0")), indent=1))
PY
grep -A3 "backward\|linear_w8a8\|bmm" gpurun_out/r06/host_us_call11.txt | head -30
timeout 900 python tools/wq_k_sweep.py 16384 4096 2>&1 | grep -v amdgpu > gpurun_out/r06/wq_k_sweep.txt
cat gpurun_out/r06/wq_k_sweep.txt
