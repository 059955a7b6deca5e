#!/bin/bash
# round 6, call 6: GPU suite with the tuned plan of the 128-column tiles; the plan's own choice at 17 .. 512 rows against the 256-row tiles and the vendor GEMM
mkdir -p gpurun_out/r06
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r06/gputests_call6.txt
timeout 1500 python tools/wq_rows_sweep.py 2>&1 | grep -v amdgpu > gpurun_out/r06/wq_rows_sweep.txt
tail -6 gpurun_out/r06/gputests_call6.txt; tail -12 gpurun_out/r06/wq_rows_sweep.txt | cut -c1-250
