#!/bin/bash
# ThreadSanitizer and Address/UB-Sanitizer runs of the host-side solver code (no GPU, no HIP): tools/sanitize/run.sh [tsan|asan|all]
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); OUT=${TMPDIR:-/tmp}/dsss_sanitize; mkdir -p "$OUT"
MODE=${1:-all}
SRC="$HERE/pg_sym_harness.cpp $ROOT/diasss_amd/csrc/dsss_pg_sym.cpp"
if [ "$MODE" = tsan ] || [ "$MODE" = all ]; then
  g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -w -I"$ROOT/diasss_amd/csrc" $SRC -o "$OUT/h_tsan"
  TSAN_OPTIONS="halt_on_error=1 exitcode=66" "$OUT/h_tsan"
fi
if [ "$MODE" = asan ] || [ "$MODE" = all ]; then
  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread -w -I"$ROOT/diasss_amd/csrc" $SRC -o "$OUT/h_asan"
  "$OUT/h_asan"
fi
echo "sanitizers: clean"
