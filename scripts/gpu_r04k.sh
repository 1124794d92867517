#!/bin/bash
# the call pattern in enkiTS's own order (threads x (threads - 1) partitions; the owning thread pops from the end, the others steal from the front)
T=${1:-r04k}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads || exit 1
{ /tmp/exp_threads 100002 1000; VMD_SDF=1 /tmp/exp_threads 100002 10000; /tmp/exp_threads 1000002 200; echo "## read-ahead off"; VMD_OPTS="readahead=0" /tmp/exp_threads 100002 1000; } 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
g++ -std=c++17 -O2 tests/native/stress_readahead.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/stress_ra && timeout 600 /tmp/stress_ra 400 240 30000 31 2>&1 | tail -1
