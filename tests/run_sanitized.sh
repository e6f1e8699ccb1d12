#!/bin/bash
# The test suites against sanitizer builds of the host code (planner, runtime, host mixer; device code as shipped):
#   bash tests/run_sanitized.sh ubsan [pytest args]     UBSan + libstdc++ assertions; works with and without a GPU
#   bash tests/run_sanitized.sh asan  [pytest args]     AddressSanitizer; CPU suite (dry-run mixer and planner)
# Default pytest args: tests -x -q -m "not gpu". Round 2: both green on the 425 CPU tests; ubsan also on the 155 GPU
# tests and 45 000 random scenes of tests/gpu_long_fuzz.py on an MI355X.
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
KIND=${1:-ubsan}; shift || true
make -s -C "$HERE/server_amd/csrc" sanitize
export CGHIP_LIB_DIR="$HERE/server_amd/csrc/build/san/$KIND"
if [ "$KIND" = asan ]; then
    # the interpreter is not instrumented: the runtime is preloaded (after whatever is preloaded already), and libstdc++
    # with it so that the runtime finds __cxa_throw when it starts
    export LD_PRELOAD="${LD_PRELOAD:+$LD_PRELOAD:}$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libstdc++.so)"
    export ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0:halt_on_error=1
else
    export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
fi
cd "$HERE"
if [ $# -eq 0 ]; then set -- tests -x -q -m "not gpu"; fi
exec python -m pytest "$@"
