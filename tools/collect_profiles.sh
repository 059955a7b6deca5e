#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats of the default bench command, two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains), the same two over tools/hbm_probe.py, the plain bench line.
# EVERY profiled command runs under `timeout` (a profiled process that aborts can sit in rocprofv3's signal handler for ever).
# Raw outputs are summarised in place and deleted (gpurun copies back at most 64 MiB).
set -u
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/profiles
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o bench -- python3 bench.py > $OUT/${TAG}_bench_under_rocprof.log 2>&1
echo "trace rc=$?"
python3 tools/rocprof_summary.py $OUT/bench_trace/bench_results.db $OUT/${TAG}_bench_kernel_stats.md "$TAG — rocprofv3 --kernel-trace --stats -- python3 bench.py (Llama-3-8B W8A8 forward, B=8 S=2048, 1x MI355X)"
rm -rf $OUT/bench_trace
grep '^{' $OUT/${TAG}_bench_under_rocprof.log > $OUT/${TAG}_bench_line_under_rocprof.json
rm -f $OUT/${TAG}_bench_under_rocprof.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o pmc --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-graph --calib-seqs 8 --no-side-measurements > $OUT/pmc_$c.log 2>&1
  echo "$c rc=$?"
  python3 tools/pmc_summary.py $(ls $OUT/pmc_$c/*counter_collection.csv | head -1) $c $OUT/${TAG}_pmc_${c}.json
  rm -rf $OUT/pmc_$c $OUT/pmc_$c.log
done
# the HBM kernels of bench.py's hbm_kernels table, same two counters (their launches are in the side measurements)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmcm_$c -o pmc --output-format csv -- python3 tools/hbm_probe.py > $OUT/pmcm_$c.log 2>&1
  echo "hbm $c rc=$?"
  python3 tools/pmc_summary.py $(ls $OUT/pmcm_$c/*counter_collection.csv | head -1) $c $OUT/${TAG}_pmc_hbm_${c}.json
  rm -rf $OUT/pmcm_$c $OUT/pmcm_$c.log
done
# the plain line looks its traffic figures up in profiles/: the passes above are the ones of THESE kernel sources
cp $OUT/${TAG}_pmc_*.json profiles/ 2>/dev/null
timeout 400 python3 bench.py > $OUT/${TAG}_bench_plain.log 2>&1
grep '^{' $OUT/${TAG}_bench_plain.log > $OUT/${TAG}_bench_line.json
rm -f $OUT/${TAG}_bench_plain.log
du -sh $OUT; ls -la $OUT
