#!/bin/bash
# Kernel time per forward step: rocprofv3 --kernel-trace --stats of bench.py with 8 calibration sequences, 40 timed steps and
# no side measurements, so that the forward dominates the trace. -> gpurun_out/step_trace/ (+ .md summary by tools/summarize_rocprof.py)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out/step_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/step_trace -o step -- python3 bench.py --steps 40 --warmup 3 --calib-seqs 8 --no-side-measurements > gpurun_out/step_trace/bench.json 2> gpurun_out/step_trace/err.log
tail -1 gpurun_out/step_trace/bench.json | cut -c1-200
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/step_trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"]):7d} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
