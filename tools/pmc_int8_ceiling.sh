#!/bin/bash
# Where the int8 GEMM stands against what the matrix pipe sustains on toggling operands (VERDICT r4 item 1's alternative):
# matrix-pipe busy cycles, GPU-active cycles (-> effective clock = GRBM_GUI_ACTIVE / duration) and memory-side counters of
#   ours    w8a8_gemm256fq_kernel on the gate/up and down_proj shapes (tools/gemm_probe.py, uniform random codes)
#   vendor  torch._int_mm's Cijk_..I8II.. kernel on the same shapes (tools/int8_vendor_probe.py)
#   probe   tools/probes/mfma_power: back-to-back v_mfma_i32_16x16x64_i8 / 32x32x32 on toggling operands, no memory traffic
# Separate --pmc passes, never combined with other trace domains; every profiled command under `timeout`.
# -> gpurun_out/r05/int8_ceiling_raw.txt (summarised by tools/int8_ceiling_md.py)
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05; mkdir -p $OUT
RAW=$OUT/int8_ceiling_raw.txt; : > $RAW
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/probes/mfma_power.hip || exit 1
i=0
run_pass() {  # label, counters..., then "--" and the command
  local label=$1; shift
  local counters=()
  while [ "$1" != "--" ]; do counters+=("$1"); shift; done
  shift
  i=$((i+1)); rm -rf /tmp/ic_$i
  timeout 300 rocprofv3 --kernel-trace --pmc "${counters[@]}" -d /tmp/ic_$i -o pmc --output-format csv -- "$@" > /tmp/ic_$i.log 2>&1
  echo "== $label | ${counters[*]} | rc=$?" >> $RAW
  python3 - $i <<'PY' >> $RAW
import csv, glob, sys, collections
i = sys.argv[1]
cc = glob.glob(f"/tmp/ic_{i}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"/tmp/ic_{i}/**/*kernel_trace.csv", recursive=True)
def key(name):
    if "w8a8_gemm256fq" in name: return "ours"
    if "Cijk" in name: return "vendor"
    if "probe<" in name: return name.split("(")[0].replace("void ", "")
    return None
dur = collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt[0])):
        k = key(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
if cc:
    for r in csv.DictReader(open(cc[0])):
        k = key(r["Kernel_Name"])
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(set(dur) | set(agg)):
    d = dur.get(k, [])
    d = d[1:] if len(d) > 1 else d
    line = f"  {k}: launches {len(dur.get(k, []))} avg_us {sum(d) / max(1, len(d)):.1f}"
    for c, v in sorted(agg.get(k, {}).items()):
        v = v[1:] if len(v) > 1 else v
        line += f" | {c} {sum(v) / max(1, len(v)):.0f}"
    print(line)
if not cc: print("  no counter file:", open(f"/tmp/ic_{i}.log").read()[-400:].replace("\n", " "))
PY
  rm -rf /tmp/ic_$i
}
for shape in "16384 14336 4096" "16384 4096 14336"; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    run_pass "ours $shape" $set -- python3 tools/gemm_probe.py $shape 4
    run_pass "vendor $shape" $set -- python3 tools/int8_vendor_probe.py $shape 4
  done
done
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  run_pass "probe" $set -- /tmp/mfma_power
done
timeout 120 /tmp/mfma_power >> $RAW 2>&1
cat $RAW | tail -60
