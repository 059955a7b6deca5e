#!/bin/bash
# Board power and clocks while the headline forward replays (tools/rope_ab.py): rocm-smi sampled twice a second beside it.
# usage (GPU box, repo root): bash tools/power_trace.sh [out.txt]
OUT=${1:-gpurun_out/power_trace.txt}
mkdir -p "$(dirname "$OUT")"
python3 tools/rope_ab.py 120 > "$OUT.run" 2>&1 &
PID=$!
: > "$OUT"
while kill -0 $PID 2>/dev/null; do
  echo "== $(date +%s.%N)" >> "$OUT"
  rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use|Temperature \(Sensor (edge|junction|memory)" >> "$OUT"
  sleep 0.5
done
wait $PID
grep -v amdgpu "$OUT.run" >> "$OUT"
rm -f "$OUT.run"
