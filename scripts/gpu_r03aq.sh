#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r03aq
for a in "c2 1" "c2 16" "c4 1"; do timeout 300 python scripts/exp_round_cost.py $a 2>/dev/null | tee -a gpurun_out/r03aq/round_cost.txt; done
