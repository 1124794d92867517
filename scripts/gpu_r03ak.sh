#!/bin/bash
# r03ak: the tree at the end of round 3: GPU suite, smoke, the default bench line, 1 000 evaluator life cycles on the resource cache
T=${1:-r03ak}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
bash scripts/gpu_suite.sh $T
g++ -std=c++17 -O2 tests/native/stress_eval.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/stress_eval || exit 1
SECONDS=0
HSA_ENABLE_SDMA=1 AMD_LOG_LEVEL=1 timeout 900 /tmp/stress_eval 1000 48 > $O/stress.out 2> $O/stress.err; rc=$?
echo "stress rc=$rc $(tail -1 $O/stress.out) wall=${SECONDS}s stderr lines: $(wc -l < $O/stress.err)" | tee $O/stress.txt
