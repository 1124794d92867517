#!/bin/bash
T=${1:-r03ai}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for a in "c2 device" "c2 xtc" "c4 device" "c3 device" "c5 device"; do timeout 600 python scripts/exp_fresh_eval.py $a 2>> $O/err.log | tee -a $O/fresh.txt; done
grep -v amdgpu.ids $O/err.log | tail -5
