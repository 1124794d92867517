#!/bin/bash
T=${1:-r03al}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for a in c2 c4 c3; do timeout 600 python scripts/exp_threads.py $a 2>> $O/err.log | tee -a $O/threads.txt; done
grep -v amdgpu.ids $O/err.log | tail -5
