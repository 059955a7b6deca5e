#!/bin/bash
# LDS-side counters of the bf16-image weight-only GEMM against the vendor's bf16 GEMM (gate/up shape, T = 16384):
# tools/pmc_wq_lds.sh -> gpurun_out/pmc_wq_lds.txt. Every profiled command under `timeout`.
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_wq_lds.txt; : > $OUT
i=0
for shape in "16384 14336 4096"; do
for set in "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1)); rm -rf /tmp/pl_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pl_$i -o pmc --output-format csv -- python3 tools/wq_probe.py $shape 4 > /tmp/pl_$i.log 2>&1
  echo "== shape $shape pass $i rc=$? : $set" >> $OUT
  python3 - $i <<'PY' >> $OUT
import csv, glob, sys, collections
i = sys.argv[1]
cc = glob.glob(f"/tmp/pl_{i}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"/tmp/pl_{i}/**/*kernel_trace.csv", recursive=True)
def key(name):
    if "wq_gemm256_kernel" in name: return "ours"
    if "Cijk" in name: return "vendor"
    return None
if kt:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        k = key(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        print(f"  {k}: {sum(v[1:]) / max(1, len(v) - 1):.1f} us per launch ({len(v)} launches)")
if cc:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        k = key(r["Kernel_Name"])
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c in sorted({c for k in agg for c in agg[k]}):
        o = agg["ours"].get(c, [0]); v = agg["vendor"].get(c, [0])
        mo, mv = sum(o) / len(o), sum(v) / len(v)
        print(f"   {c:30s} ours {mo:16.0f}   vendor {mv:16.0f}   ours/vendor {mo / max(mv, 1):.3f}")
else:
    print("  no counter file:", open(f"/tmp/pl_{i}.log").read()[-600:])
PY
done
done
cat $OUT
