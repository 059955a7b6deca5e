#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root: micro table (rocprofv3 kernel trace + CPU eager chain), the
# forward-step kernel stats, the weight-only GEMM timings and the BASELINE configs 2-5. Outputs under gpurun_out/r03/.
# EVERY profiled command runs under `timeout`: a profiled process that aborts can otherwise sit in rocprofv3's signal handler.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03
mkdir -p $OUT
WHAT=${1:-all}
if [ $WHAT = all ] || [ $WHAT = micro ]; then
  rm -rf $OUT/micro_trace
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/micro_trace -o micro -- python3 tools/micro_table.py --probe $OUT/micro_plan.json > $OUT/micro_probe.log 2>&1
  echo "micro trace rc=$?"
  [ -f $OUT/micro_cpu.json ] || timeout 600 python3 tools/micro_table.py --cpu $OUT/micro_cpu.json > $OUT/micro_cpu.log 2>&1
  python3 tools/micro_table.py --merge $(ls $OUT/micro_trace/*/*kernel_trace.csv $OUT/micro_trace/*kernel_trace.csv 2>/dev/null | head -1) $OUT/micro_plan.json $OUT/micro_cpu.json $OUT/r03_micro.md > /dev/null
  rm -rf $OUT/micro_trace
fi
if [ $WHAT = all ] || [ $WHAT = step ]; then
  rm -rf $OUT/step_trace
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/step_trace -o step -- python3 bench.py --steps 10 --warmup 2 --calib-seqs 8 --no-side-measurements > $OUT/step_bench.json 2> $OUT/step_err.log
  echo "step trace rc=$?"
  python3 tools/rocprof_summary.py $OUT/step_trace/step_results.db $OUT/r03_forward_step_kernel_stats.md "r03 — rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --calib-seqs 8 --no-side-measurements (the forward dominates: 13 forwards + one 8-sequence calibration step)"
  rm -rf $OUT/step_trace
fi
if [ $WHAT = all ] || [ $WHAT = wq ]; then
  timeout 300 python3 tools/wq_time.py 16384 > $OUT/r03_wq_time.txt 2>&1
  timeout 300 python3 tools/wq_time.py 4096 >> $OUT/r03_wq_time.txt 2>&1
  timeout 300 python3 tools/wq_time.py 2048 >> $OUT/r03_wq_time.txt 2>&1
fi
if [ $WHAT = all ] || [ $WHAT = configs ]; then
  timeout 1500 python3 tools/bench_configs.py --out $OUT/r03_configs.json > $OUT/configs.log 2>&1
fi
du -sh $OUT; ls $OUT
