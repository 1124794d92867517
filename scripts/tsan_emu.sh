#!/bin/bash
# Host-side data-race check: the emulator build (same .hip/.cpp sources, g++) and the C++ pool-thread programs of tests/native
# under ThreadSanitizer.  The emulator's fibers are announced to TSan (tests/emu/emu.cpp: __tsan_create_fiber /
# __tsan_switch_to_fiber), so what is checked is the host code around the launches: the read-ahead's lock-free request marks and
# flight word, the combining queue, the block commits, interrupt / clear / free against running calls, the shim's registry, merges of several ranks driven from threads of one process.
# usage: bash scripts/tsan_emu.sh [out_dir]      (about 45 minutes on 8 cores: TSan costs 10 - 20 x)
set -e
cd "$(dirname "$0")/.."
R=$(pwd)
OUT=${1:-/tmp/viamd_tsan}; mkdir -p $OUT
LIB=$(VIAMD_EMU_SANITIZE=thread python -c "import sys; sys.path.insert(0, 'tests'); import conftest; print(conftest.build_emu())" 2>/dev/null | tail -1)
export TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:history_size=4"
build() {   # src exe [extra include]
  g++ -std=c++20 -O1 -g -fsanitize=thread -Wno-format "$1" -I$R/include -I$R/oracle ${3:+-I$3} $LIB -Wl,-rpath,$(dirname $LIB) -lpthread -o "$2"
}
# VIAMD's own call sites (verbatim slices of /root/reference/src, tests/native/ref_callsites.cpp): the reference's main-loop block polls and
# re-creates evals while its pool tasks run inside md_script_eval_frame_range - the shim's epoch / record refresh / settle callback under TSan
python - <<PY
import subprocess, sys
sys.path.insert(0, "$R")
from oracle import make_ref
if make_ref.available():
    make_ref.write_callsite_slices()
    for exe, extra in (("$OUT/ref_callsites", []), ("$OUT/ref_callsites_deferred", ["-DVMD_SHIM_DEFERRED_SETTLE"])):
        subprocess.check_call(make_ref.callsites_compile_cmd(exe, ["-g", "-fsanitize=thread"] + extra + ["$LIB", "-Wl,-rpath,$(dirname $LIB)"], opt="-O1"))
PY
build tests/native/stress_readahead.cpp $OUT/stress_ra
build tests/native/stress_eval.cpp $OUT/stress_eval
build tests/native/shim_callsites.cpp $OUT/shim_callsites $R/tests/native
build tests/native/shim_default_script.cpp $OUT/shim_default_script $R/tests/native
build tests/native/exp_threads.cpp $OUT/exp_threads
build tests/native/reduce_threads.cpp $OUT/reduce_threads
build tests/native/concurrent_evals.cpp $OUT/concurrent_evals
cd $OUT
# the programs run side by side (the emulator executes one launch at a time per process, so each is about one core)
run() {   # name args...      (TSAN_ONLY="name name ..." runs only those)
  local name=$1; shift
  if [ -n "$TSAN_ONLY" ] && [[ " $TSAN_ONLY " != *" $name "* ]]; then return; fi
  ( t0=$(date +%s)
    prc=0; timeout 5400 "$@" > $OUT/$name.log 2>&1 || prc=$?      # (set -e: a failing program must still leave its .result)
    n=$(grep -c "WARNING: ThreadSanitizer" $OUT/$name.log || true)
    echo "$name: rc $prc, $(($(date +%s) - t0)) s, ThreadSanitizer warnings: $n, last line: $(grep -v '^$' $OUT/$name.log | tail -1 | cut -c1-200)" > $OUT/$name.result
    [ "$n" = "0" ] || grep "SUMMARY" $OUT/$name.log | sort | uniq -c >> $OUT/$name.result ) &
}
run reduce_threads $OUT/reduce_threads 3 12 600
run concurrent_evals $OUT/concurrent_evals 2 10 600 $OUT
run shim_callsites $OUT/shim_callsites 12
run shim_default_script $OUT/shim_default_script 12
# (the reference's own readers use plain loads by design: scripts/tsan_reference.supp names them - and only them)
if [ -x $OUT/ref_callsites ]; then
  # history_size=7: more of the reference's reader stacks can be restored (those that cannot are recognised by the writer, see the .supp file)
  REF_TSAN="${TSAN_OPTIONS/history_size=4/history_size=7}:suppressions=$R/scripts/tsan_reference.supp"
  TSAN_OPTIONS="$REF_TSAN" run ref_callsites $OUT/ref_callsites 12 $OUT
  TSAN_OPTIONS="$REF_TSAN" run ref_callsites_deferred $OUT/ref_callsites_deferred 12 $OUT
fi
run stress_eval $OUT/stress_eval 2 6
run exp_threads_rdf $OUT/exp_threads ${TSAN_EXP_ARGS:-900 64}
run stress_ra_sdf $OUT/stress_ra 3 16 600 9 sdf
run stress_ra $OUT/stress_ra ${TSAN_RA_ARGS:-8 40 900 7}
wait
cat $OUT/*.result
! grep -q "warnings: [1-9]\|rc [1-9]" $OUT/*.result
