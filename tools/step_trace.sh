#!/bin/bash
# Kernel time per forward step: rocprofv3 --kernel-trace --stats of bench.py with 8 calibration sequences, 40 timed steps and
# no side measurements, so that the forward dominates the trace. -> gpurun_out/step_trace/ (+ .md summary by tools/summarize_rocprof.py)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out/step_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/step_trace -o step -- python3 bench.py --steps 40 --warmup 3 --calib-seqs 8 --no-side-measurements > gpurun_out/step_trace/bench.json 2> gpurun_out/step_trace/err.log
tail -1 gpurun_out/step_trace/bench.json | cut -c1-200
python3 tools/rocprof_summary.py gpurun_out/step_trace/step_results.db gpurun_out/step_trace/forward_step_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 40 --warmup 3 --calib-seqs 8 --no-side-measurements (the forward dominates: 40 timed steps)"
rm -f gpurun_out/step_trace/step_results.db
head -24 gpurun_out/step_trace/forward_step_kernel_stats.md | cut -c1-160
