#!/bin/bash
# SQ / GRBM counter passes over one W8A8 GEMM shape (run on the GPU box from the repo root via gpurun).
# usage: tools/pmc_gemm.sh M N K TAG   -> gpurun_out/pmc_gemm/TAG_*.csv (one csv per pass, GEMM rows only)
set -u
export TMPDIR=/tmp
M=${1:-16384}; N=${2:-14336}; K=${3:-4096}; TAG=${4:-gateup}
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT
pass() {
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$name -o pmc --output-format csv -- python3 tools/gemm_probe.py $M $N $K 3 > /tmp/pmc_$name.log 2>&1
  echo "$name rc=$?"
  f=$(ls /tmp/pmc_$name/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then (head -1 $f; grep w8a8_gemm $f) > $OUT/${TAG}_$name.csv; else tail -5 /tmp/pmc_$name.log; fi
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC
pass tcc TCC_HIT_sum TCC_MISS_sum
ls -la $OUT
