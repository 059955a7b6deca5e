#!/bin/bash
# SQ / GRBM counter passes over the attention launch (run on the GPU box from the repo root via gpurun).
# -> gpurun_out/pmc_attn/TAG_*.csv (one csv per pass, attention rows only) and a summary table on stdout
set -u
export TMPDIR=/tmp
TAG=${1:-attn}
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
pass() {
  local name=$1; shift
  rm -rf /tmp/pmca_$name
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmca_$name -o pmc --output-format csv -- python3 tools/attn_once.py > /tmp/pmca_$name.log 2>&1
  echo "$name rc=$?"
  f=$(ls /tmp/pmca_$name/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep attention_fwd $f) > $OUT/${TAG}_$name.csv; else tail -5 /tmp/pmca_$name.log; fi
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC
pass sq3 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
python3 tools/pmc_table.py $OUT/${TAG}_sq1.csv $OUT/${TAG}_sq2.csv $OUT/${TAG}_sq3.csv
