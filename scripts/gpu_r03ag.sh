#!/bin/bash
T=${1:-r03ag}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python scripts/fuzz_xtc.py 300 11 gpu 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|^ *|" | tail -2 | cut -c1-500 | tee $O/fuzz_xtc_seed11.txt
