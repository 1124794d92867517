#!/bin/bash
# read-ahead after the contention pass (one flight word, striped request marks, no shared counters on the fast path)
T=${1:-r04c}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads || exit 1
{
echo "== read-ahead on (default)"
/tmp/exp_threads 100002 1000; VMD_SDF=1 /tmp/exp_threads 100002 10000; /tmp/exp_threads 1000002 200
echo "== blocks of 512 / 1024 for the sdf"
VMD_OPTS="readahead_block=512" VMD_SDF=1 /tmp/exp_threads 100002 10000
VMD_OPTS="readahead_block=1024" VMD_SDF=1 /tmp/exp_threads 100002 10000
echo "== first region 512"
VMD_OPTS="readahead_frames=512" /tmp/exp_threads 100002 1000; VMD_OPTS="readahead_frames=512" VMD_SDF=1 /tmp/exp_threads 100002 10000
} 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "read_ahead or pool_threads or filtered or concurrently or sharding" > $O/pytest_ra.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ra.log
