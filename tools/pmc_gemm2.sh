#!/bin/bash
# SQ counters of the W8A8 GEMM variants on one shape: tools/pmc_gemm2.sh M N K  (base, then FFQ_GEMM_4W=1)
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
M=${1:-16384}; N=${2:-4096}; K=${3:-14336}
for mode in base 4w; do
  if [ $mode = 4w ]; then export FFQ_GEMM_4W=1; else unset FFQ_GEMM_4W; fi
  for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"; do
    rm -rf /tmp/pg; rocprofv3 --kernel-trace --pmc $pass -d /tmp/pg -o pmc --output-format csv -- python3 tools/gemm_probe.py $M $N $K 3 > /tmp/pg.log 2>&1
    python3 - "$mode" <<'PY'
import csv, glob, sys, collections
mode = sys.argv[1]
cc = glob.glob("/tmp/pg/**/*counter_collection.csv", recursive=True)
kt = glob.glob("/tmp/pg/**/*kernel_trace.csv", recursive=True)
dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if "w8a8_gemm" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc[0])):
    if "w8a8_gemm" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{mode}: {sum(dur)/len(dur):.1f} us  " + "  ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(agg.items())))
PY
  done
done
