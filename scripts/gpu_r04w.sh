#!/bin/bash
# last randomised campaign of round 4 on the final code: scenario seeds not used before (the single-call path and VIAMD's pool pattern)
T=${1:-r04w}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
timeout 100 python scripts/fuzz_gpu.py 60000 60700 2>&1 | tail -2
timeout 110 python scripts/fuzz_pool.py 61000 61500 gpu 2>&1 | tail -2
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
