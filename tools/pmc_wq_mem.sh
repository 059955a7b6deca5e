#!/bin/bash
# Memory-side counters (L2 hits / misses, fabric bytes) of the bf16-image weight-only GEMM against the vendor bf16 GEMM:
# tools/pmc_wq_mem.sh -> gpurun_out/pmc_wq_mem.txt. Every profiled command under `timeout`.
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_wq_mem.txt; : > $OUT
i=0
for shape in "16384 14336 4096" "16384 4096 14336"; do
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1)); rm -rf /tmp/pl_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pl_$i -o pmc --output-format csv -- python3 tools/wq_probe.py $shape 4 > /tmp/pl_$i.log 2>&1
  echo "== shape $shape pass $i rc=$? : $set" >> $OUT
  python3 - $i <<'PY' >> $OUT
import csv, glob, sys, collections
i = sys.argv[1]
cc = glob.glob(f"/tmp/pl_{i}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"/tmp/pl_{i}/**/*kernel_trace.csv", recursive=True)
def key(name):
    if "wq_gemm256_kernel" in name or "wq_gemm4w_kernel" in name: return "ours"
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
