#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pool_threads or life_cycles or sharding_threads" 2>&1 | tail -3
