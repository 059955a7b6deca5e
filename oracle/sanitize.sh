#!/bin/bash
# The checker checked: build the C oracle with AddressSanitizer + UndefinedBehaviorSanitizer and run the CPU suite's oracle-driven
# tests on it (GPU sanitizers are not available on this pool; this covers the restatement every parity claim rests on).
#   bash oracle/sanitize.sh [pytest args]        (default: the golden-fixture, host-logic and Llama-harness tests)
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/ffq_oracle_san; mkdir -p $OUT
gcc -O1 -g -std=c11 -fPIC -Wall -Wextra -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o $OUT/libffq_oracle.so oracle/ffq_oracle.c -lm
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export FFQ_ORACLE_SO=$OUT/libffq_oracle.so
if [ $# -gt 0 ]; then exec python -m pytest "$@"; fi
exec python -m pytest tests/test_oracle_golden.py tests/test_host_logic.py tests/test_llama_harness.py tests/test_eager_chain.py -x -q -m "not gpu"
