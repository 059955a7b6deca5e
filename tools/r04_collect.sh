#!/bin/bash
# Round-4 evidence, run on the GPU box from the repo root. Outputs under gpurun_out/r04/ (copy what is judged into profiles/).
#   micro    BASELINE 3's A1 / A2 / A4 table from a rocprofv3 kernel trace beside the CPU eager chain
#   step     kernel stats of the forward step (bench.py, 10 timed steps, 8 calibration sequences)
#   wq       weight-only GEMM timings at 16384 / 4096 / 2048 / 512 / 64 / 1 tokens + the split sweep
#   traces   rocprofv3 kernel traces of one cfg2 and one cfg4 forward at B=1 and B=8 (which GEMMs the decoder linears run)
#   configs  BASELINE configs 2-5 (tools/bench_configs.py)
# EVERY profiled command runs under `timeout`: a profiled process that aborts can otherwise sit in rocprofv3's signal handler.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r04
mkdir -p $OUT
WHAT=${1:-all}
if [ $WHAT = all ] || [ $WHAT = micro ]; then
  rm -rf $OUT/micro_trace
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/micro_trace -o micro -- python3 tools/micro_table.py --probe $OUT/micro_plan.json > $OUT/micro_probe.log 2>&1
  echo "micro trace rc=$?"
  [ -f $OUT/micro_cpu.json ] || timeout 600 python3 tools/micro_table.py --cpu $OUT/micro_cpu.json > $OUT/micro_cpu.log 2>&1
  python3 tools/micro_table.py --merge $(ls $OUT/micro_trace/*/*kernel_trace.csv $OUT/micro_trace/*kernel_trace.csv 2>/dev/null | head -1) $OUT/micro_plan.json $OUT/micro_cpu.json $OUT/r04_micro.md > /dev/null
  rm -rf $OUT/micro_trace
fi
if [ $WHAT = all ] || [ $WHAT = step ]; then
  rm -rf $OUT/step_trace
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/step_trace -o step -- python3 bench.py --steps 10 --warmup 2 --calib-seqs 8 --no-side-measurements > $OUT/step_bench.json 2> $OUT/step_err.log
  echo "step trace rc=$?"
  python3 tools/rocprof_summary.py $OUT/step_trace/step_results.db $OUT/r04_forward_step_kernel_stats.md "r04 — rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --calib-seqs 8 --no-side-measurements (the forward dominates: 13 forwards + one 8-sequence calibration step)"
  rm -rf $OUT/step_trace
fi
if [ $WHAT = all ] || [ $WHAT = wq ]; then
  : > $OUT/r04_wq_time.txt
  for T in 16384 4096 2048; do timeout 300 python3 tools/wq_time.py $T 2>&1 | grep -v amdgpu.ids >> $OUT/r04_wq_time.txt; done
  timeout 600 python3 tools/wq_split_sweep.py 1 64 512 2048 4096 2>&1 | grep -v amdgpu.ids > $OUT/r04_wq_split_sweep.txt
fi
if [ $WHAT = all ] || [ $WHAT = traces ]; then
  for cfg in cfg2 cfg4; do
    for b in 1 8; do
      rm -rf $OUT/trace_$cfg
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$cfg -o t -- python3 tools/cfg_trace.py $cfg $b > $OUT/trace_${cfg}_B$b.log 2>&1
      echo "$cfg B=$b trace rc=$?"
      python3 tools/rocprof_summary.py $OUT/trace_$cfg/t_results.db $OUT/r04_${cfg}_B${b}_forward_kernel_trace.md "r04 — rocprofv3 --kernel-trace --stats -- python3 tools/cfg_trace.py $cfg $b (3 forwards of B=$b, S=2048 through llama.FusedProducersForward; the weight quantizers / the one-off packing and a 256-token calibration forward included). Vendor GEMMs (Cijk_*) in this trace: lm_head only (float in the recipe, quick-start :145) — one launch per forward, 3 in all; every decoder linear is wq_gemm256_kernel or wq_gemm4w_kernel (the one-wave-per-SIMD form of its bf16-image launches)"
      rm -rf $OUT/trace_$cfg
    done
  done
fi
if [ $WHAT = all ] || [ $WHAT = configs ]; then
  timeout 2400 python3 tools/bench_configs.py --out $OUT/r04_configs.json > $OUT/configs.log 2>&1
fi
du -sh $OUT; ls $OUT
