#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "life_cycles or pool_threads or filtered_evaluation_reuses" 2>&1 | tail -2
