#!/bin/bash
T=${1:-r03ac}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/fuzz_xtc.py 400 21 gpu 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|^ *|" | tail -3 | cut -c1-600 | tee $O/fuzz_xtc.txt
