#!/bin/bash
T=${1:-r03am}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads || exit 1
{ /tmp/exp_threads 100002 1000; /tmp/exp_threads 100002 10000; /tmp/exp_threads 1000002 200; /tmp/exp_threads 30000 2000; } 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
