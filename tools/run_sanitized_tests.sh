#!/bin/bash
# CPU sanitizer pass: the oracle rebuilt with ASan + UBSan, the non-GPU test suite run against it.
# Host code only (GPU AddressSanitizer is not available on this pool).  usage: tools/run_sanitized_tests.sh [pytest args]
set -eo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$ROOT/oracle" SAN=1
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
# the sanitizer runtimes have to come first in the process; whatever is already preloaded stays
export LD_PRELOAD="$ASAN:$UBSAN${LD_PRELOAD:+:$LD_PRELOAD}"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"   # the Python interpreter's own allocations are not ours to judge
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export ORACLE_LIB="$ROOT/oracle/liboracle_san.so"
cd "$ROOT"
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
