#!/bin/bash
# Counter passes over tools/mlp_wq_probe.py: plain bf16-image weight-only GEMM against its MLP mode -> gpurun_out/pmc_mlp_wq.txt
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_mlp_wq.txt; : > $OUT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_32B_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1)); rm -rf /tmp/pm_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pm_$i -o pmc --output-format csv -- python3 tools/mlp_wq_probe.py > /tmp/pm_$i.log 2>&1
  echo "pass $i rc=$?" >> $OUT
  python3 - $i <<'PY' >> $OUT
import csv, glob, sys, collections
i = sys.argv[1]
cc = glob.glob(f"/tmp/pm_{i}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"/tmp/pm_{i}/**/*kernel_trace.csv", recursive=True)
def key(name):
    if "wq_gemm256_kernel" not in name: return None
    return "mlp" if name.rstrip(">(WLinearArgs, int) ").endswith("true") or ", true>" in name else "plain"
if kt:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt[0])):
        k = key(r["Kernel_Name"])
        if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items(): print(f"  {k}: {sum(v[1:]) / max(1, len(v) - 1):.1f} us per launch ({len(v)} launches)")
if cc:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        k = key(r["Kernel_Name"])
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for k in agg for c in agg[k]})
    for c in names:
        p = agg["plain"].get(c, [0]); m = agg["mlp"].get(c, [0])
        print(f"   {c:32s} plain {sum(p) / len(p):18.0f}   mlp {sum(m) / len(m):18.0f}   mlp / (2 plain) = {(sum(m) / len(m)) / max(1.0, 2 * sum(p) / len(p)):.3f}")
else:
    print("  no counter file; log tail:", open(f"/tmp/pm_{i}.log").read()[-600:])
PY
done
cat $OUT
