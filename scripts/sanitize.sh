#!/bin/bash
# The CPU builds under AddressSanitizer + UndefinedBehaviorSanitizer: the oracle (oracle/classpro_oracle*.c), the
# product's scalar device functions compiled for the host (tests/host_harness.cpp over classpro_amd/csrc/*.h), the
# FASTX indexer checker, the track harness and the host-only tools, driven by the whole CPU test suite.
# GPU sanitizers do not exist on this pool; this is the job that found the profile[plen] read of correct_wall_cnt
# (hazard 8).  Run from the repo root:  scripts/sanitize.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
export CP_SANITIZE=1
REP=${TMPDIR:-/tmp}/cp_sanitize.$$; mkdir -p "$REP"   # reports go to files: pytest captures stderr and an abort loses it
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:log_path=$REP/asan
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1:log_path=$REP/ubsan
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
rm -f tests/_*.srchash classpro_amd/.tools.srchash          # helpers are rebuilt with the sanitizer flags ...
set +e
LD_PRELOAD="$ASAN:$UBSAN" python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
rc=$?
# ... and the parameter-space soak of the same two implementations (K 15-63, -r 1000-60000, coverages (5,10)-(50,100))
if [ $rc -eq 0 ] && [ $# -eq 0 ]; then
  LD_PRELOAD="$ASAN:$UBSAN" python scripts/fuzz_host.py 2 | tail -1
  rc=${PIPESTATUS[0]}
fi
for f in "$REP"/*; do [ -f "$f" ] && { echo "== sanitizer report $f"; head -40 "$f"; rc=1; }; done
rm -rf "$REP"
rm -f tests/_*.srchash classpro_amd/.tools.srchash          # ... and without them by the next ordinary run
exit $rc
