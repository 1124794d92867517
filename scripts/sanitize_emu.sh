#!/bin/bash
# Host-code hygiene: the emulator build (same .hip/.cpp sources, g++) under AddressSanitizer + UBSan, driven by the CPU test
# suite.  The fibers of the emulator switch stacks with swapcontext, so stack-use-after-return detection is off.
# usage: bash scripts/sanitize_emu.sh [address,undefined|thread] [pytest args...]
set -e
cd "$(dirname "$0")/.."
SAN=${1:-address,undefined}; shift || true
export VIAMD_EMU_SANITIZE=$SAN
case "$SAN" in
  thread*) RT=$(g++ -print-file-name=libtsan.so);;
  *)       RT=$(g++ -print-file-name=libasan.so);;
esac
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:halt_on_error=1:log_path=${SAN_LOG:-/tmp/viamd_asan}
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=${SAN_LOG:-/tmp/viamd_asan}
export TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0
if [[ "$SAN" != thread* ]]; then     # the oracle (the checker) gets the same treatment
  gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -fno-fast-math -mavx2 -mfma -fopenmp -fsanitize=$SAN -fno-sanitize-recover=undefined \
      -fno-omit-frame-pointer -shared -o /tmp/libvmd_oracle_san.so oracle/vmd_oracle.c -lm
  export VMD_ORACLE_LIB=/tmp/libvmd_oracle_san.so
fi
TARGETS=("$@"); [ ${#TARGETS[@]} -eq 0 ] && TARGETS=(tests)
# libstdc++ is preloaded next to the runtime: the __cxa_throw interceptor of a preloaded libasan needs it resolvable at start-up
# (the script front-end throws and catches internally)
LD_PRELOAD="$RT $(g++ -print-file-name=libstdc++.so)" python -m pytest "${TARGETS[@]}" -q -m "not gpu" -x
