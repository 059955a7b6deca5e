#!/bin/bash
# Clock / busy counters of weight-only GEMM builds: tools/pmc_wq.sh TAG... -> gpurun_out/pmc_wq/TAG.txt (kernel time from the trace,
# GRBM_GUI_ACTIVE / SQ busy from one --pmc pass; "base" = the shipped library)
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
OUT=gpurun_out/pmc_wq; mkdir -p $OUT
for tag in "$@"; do
  rm -rf /tmp/pw_$tag
  if [ "$tag" = base ]; then unset FFQ_LIB; else export FFQ_LIB=fastforward_amd/csrc/_build/libffq_$tag.so; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d /tmp/pw_$tag -o pmc --output-format csv -- python3 tools/wq_probe.py 16384 14336 4096 4 > /tmp/pw_$tag.log 2>&1
  python3 - "$tag" <<'PY' > $OUT/$tag.txt
import csv, glob, sys, collections
tag = sys.argv[1]
cc = glob.glob(f"/tmp/pw_{tag}/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"/tmp/pw_{tag}/**/*kernel_trace.csv", recursive=True)
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt[0])):
    name = r["Kernel_Name"]
    key = "wq" if "wq_bf16" in name else ("blas" if ("Cijk" in name or "gemm" in name.lower()) else None)
    if key: dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc[0])):
    name = r["Kernel_Name"]
    key = "wq" if "wq_bf16" in name else ("blas" if ("Cijk" in name or "gemm" in name.lower()) else None)
    if key: agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in dur:
    us = sum(dur[key][1:]) / max(1, len(dur[key]) - 1)
    print(f"{tag} {key}: {us:.1f} us per launch ({len(dur[key])} launches)")
    for c, v in sorted(agg[key].items()):
        mean = sum(v) / len(v)
        extra = f"  -> {mean / 8 / us / 1e3:.3f} GHz" if c == "GRBM_GUI_ACTIVE" else ""
        print(f"   {c:28s} {mean:16.0f}{extra}")
PY
  cat $OUT/$tag.txt
done
