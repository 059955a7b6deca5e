#!/bin/bash
# Counters of the bf16-image weight-only GEMM against the vendor's bf16 GEMM on two shapes (gate/up and down_proj, T = 16384):
# tools/pmc_wq_shapes.sh -> gpurun_out/pmc_wq_shapes.txt. Every profiled command under `timeout`.
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_wq_shapes.txt; : > $OUT
i=0
for shape in "16384 14336 4096" "16384 4096 14336"; do
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
  i=$((i+1)); rm -rf /tmp/ps_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/ps_$i -o pmc --output-format csv -- python3 tools/wq_probe.py $shape 4 > /tmp/ps_$i.log 2>&1
  echo "== shape $shape pass $i rc=$?" >> $OUT
  python3 - $i <<'PY' >> $OUT
import csv, glob, sys, collections
i = sys.argv[1]
cc = glob.glob(f"/tmp/ps_{i}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"/tmp/ps_{i}/**/*kernel_trace.csv", recursive=True)
def key(name):
    if "wq_gemm256_kernel" in name: return "ours"
    if "Cijk" in name: return "vendor"
    return None
us = {}
if kt:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        k = key(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        us[k] = sum(v[1:]) / max(1, len(v) - 1)
        print(f"  {k}: {us[k]:.1f} us per launch ({len(v)} launches)")
if cc:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        k = key(r["Kernel_Name"])
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c in sorted({c for k in agg for c in agg[k]}):
        o = agg["ours"].get(c, [0]); v = agg["vendor"].get(c, [0])
        mo, mv = sum(o) / len(o), sum(v) / len(v)
        extra = f"   clocks: ours {mo / 8 / us.get('ours', 1) / 1e3:.2f} GHz vendor {mv / 8 / us.get('vendor', 1) / 1e3:.2f} GHz" if c == "GRBM_GUI_ACTIVE" and us else ""
        print(f"   {c:30s} ours {mo:16.0f}   vendor {mv:16.0f}   ours/vendor {mo / max(mv, 1):.3f}{extra}")
else:
    print("  no counter file:", open(f"/tmp/ps_{i}.log").read()[-400:])
PY
done
done
cat $OUT
