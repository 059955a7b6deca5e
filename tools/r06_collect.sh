#!/bin/bash
# Round-6 evidence, run on the GPU box from the repo root. Outputs under gpurun_out/r06/ (copy what is judged into profiles/).
#   profiles  tools/collect_profiles.sh r06: kernel stats of the default bench command, PMC FETCH_SIZE / WRITE_SIZE passes of the bench
#             and of the HBM kernels, the plain bench line
#   micro     BASELINE 3's A1 / A2 / A4 (+ A3 per row) table from a rocprofv3 kernel trace beside the CPU eager chain
#   step      kernel stats of the forward step (bench.py, 10 timed steps, 8 calibration sequences)
#   wq        weight-only GEMM timings at 16384 / 4096 / 2048 tokens, the split sweep of the 256-row-tile kernel (2048, 4096), the skinny
#             form's sweep (1 .. 16 tokens) and the library's own plan at 17 .. 512 rows (tools/wq_rows_sweep.py)
#   a3        A3 / A5 timings (one-launch and grid forms against the composed / one-block forms)
#   traces    rocprofv3 kernel traces of one cfg2 and one cfg4 forward at B=1 and B=8
#   calib     the calibration step with each estimation shortcut switched off (tools/calib_ab.py) and a calibration-dominated kernel trace
#   configs   BASELINE configs 2-5 (tools/bench_configs.py)
# EVERY profiled command runs under `timeout`: a profiled process that aborts can otherwise sit in rocprofv3's signal handler.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06
mkdir -p $OUT
WHAT=${1:-all}
if [ $WHAT = all ] || [ $WHAT = profiles ]; then
  bash tools/collect_profiles.sh r06 > $OUT/collect_profiles.log 2>&1
  cp gpurun_out/profiles/r06_* $OUT/ 2>/dev/null
fi
if [ $WHAT = all ] || [ $WHAT = micro ]; then
  rm -rf $OUT/micro_trace
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/micro_trace -o micro -- python3 tools/micro_table.py --probe $OUT/micro_plan.json > $OUT/micro_probe.log 2>&1
  echo "micro trace rc=$?"
  [ -f $OUT/micro_cpu.json ] || timeout 900 python3 tools/micro_table.py --cpu $OUT/micro_cpu.json > $OUT/micro_cpu.log 2>&1
  python3 tools/micro_table.py --merge $(ls $OUT/micro_trace/*/*kernel_trace.csv $OUT/micro_trace/*kernel_trace.csv 2>/dev/null | head -1) $OUT/micro_plan.json $OUT/micro_cpu.json $OUT/r06_micro.md > /dev/null
  rm -rf $OUT/micro_trace
fi
if [ $WHAT = all ] || [ $WHAT = step ]; then
  rm -rf $OUT/step_trace
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/step_trace -o step -- python3 bench.py --steps 10 --warmup 2 --calib-seqs 8 --no-side-measurements > $OUT/step_bench.json 2> $OUT/step_err.log
  echo "step trace rc=$?"
  python3 tools/rocprof_summary.py $OUT/step_trace/step_results.db $OUT/r06_forward_step_kernel_stats.md "r06 — rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --calib-seqs 8 --no-side-measurements (the forward dominates: 13 forwards + one 8-sequence calibration step)"
  rm -rf $OUT/step_trace
fi
if [ $WHAT = all ] || [ $WHAT = wq ]; then
  : > $OUT/r06_wq_time.txt
  for T in 16384 4096 2048; do timeout 300 python3 tools/wq_time.py $T 2>&1 | grep -v amdgpu.ids >> $OUT/r06_wq_time.txt; done
  timeout 600 python3 tools/wq_split_sweep.py 2048 4096 2>&1 | grep -v amdgpu.ids > $OUT/r06_wq_split_sweep.txt
  timeout 900 python3 tools/wq_skinny_sweep.py 1 4 8 16 2>&1 | grep -v amdgpu.ids > $OUT/r06_wq_skinny_sweep.txt
  timeout 1200 python3 tools/wq_rows_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_wq_rows_sweep_final.txt
fi
if [ $WHAT = all ] || [ $WHAT = a3 ]; then
  timeout 300 python3 tools/a3_time.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_a3_a5_time.txt
fi
if [ $WHAT = all ] || [ $WHAT = traces ]; then
  for cfg in cfg2 cfg4; do
    for b in 1 8; do
      rm -rf $OUT/trace_$cfg
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$cfg -o t -- python3 tools/cfg_trace.py $cfg $b > $OUT/trace_${cfg}_B$b.log 2>&1
      echo "$cfg B=$b trace rc=$?"
      python3 tools/rocprof_summary.py $OUT/trace_$cfg/t_results.db $OUT/r06_${cfg}_B${b}_forward_kernel_trace.md "r06 — rocprofv3 --kernel-trace --stats -- python3 tools/cfg_trace.py $cfg $b (3 forwards of B=$b, S=2048 through llama.FusedProducersForward; the weight quantizers / the one-off packing and a 256-token calibration forward included). Vendor GEMMs (Cijk_*) in this trace: lm_head only (float in the recipe, quick-start :145) — one launch per forward, 3 in all; every decoder linear is wq_gemm256_kernel, wq_gemm4w_kernel (the one-wave-per-SIMD form of its bf16-image launches) or, up to 512 rows, wq_mid_* / wq_skinny_*"
      rm -rf $OUT/trace_$cfg
    done
  done
fi
if [ $WHAT = all ] || [ $WHAT = calib ]; then
  timeout 900 python3 tools/calib_ab.py 32 8 2>&1 | grep -v "amdgpu\|Warning\|_warn_once" > $OUT/r06_calib_ab_final.txt
  rm -rf $OUT/calib_trace
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/calib_trace -o calib -- python3 bench.py --steps 1 --warmup 0 --calib-seqs 64 --no-side-measurements > $OUT/calib_bench.json 2> $OUT/calib_err.log
  python3 tools/rocprof_summary.py $OUT/calib_trace/calib_results.db $OUT/r06_calibration_kernel_stats.md "r06 — rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --calib-seqs 64 --no-side-measurements (calibration dominates: 1 + 8 calibration steps of 8 sequences, 3 forwards)"
  rm -rf $OUT/calib_trace
fi
if [ $WHAT = all ] || [ $WHAT = configs ]; then
  timeout 2400 python3 tools/bench_configs.py --out $OUT/r06_configs.json > $OUT/configs.log 2>&1
fi
du -sh $OUT; ls $OUT
