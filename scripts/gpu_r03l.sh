#!/bin/bash
# r03l: the two GPU tests added after r03k (torch fallback collective on device memory; inline-asm vs C++ twin on gfx950)
T=${1:-r03l}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_zz_late_gpu.py -m gpu -x -q > $O/pytest_late.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_late.log
