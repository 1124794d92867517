#!/bin/bash
# r03an: frame_range as VIAMD calls it (pool threads, ranges of 1 - 16 frames, all on one eval): gather window + lazy views, A/B
T=${1:-r03an}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads || exit 1
{ for o in "gather_us=0 lazy_views=0" "gather_us=150 lazy_views=1"; do
    echo "## $o"
    VMD_OPTS="$o" /tmp/exp_threads 100002 1000; VMD_OPTS="$o" /tmp/exp_threads 1000002 200; VMD_OPTS="$o" /tmp/exp_threads 30000 2000; VMD_SDF=1 VMD_OPTS="$o" /tmp/exp_threads 100002 10000
  done; } 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
timeout 900 python -m pytest tests/test_native.py tests/test_gpu_parity.py -m gpu -x -q -k "native or sharding or concurrently or interrupt or running_source" 2>&1 | tail -3
