#!/bin/bash
# read-ahead under VIAMD's call pattern: exp_threads with and without, the new GPU test, option sweep
T=${1:-r04b}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads || exit 1
{
echo "== read-ahead on (default)"
/tmp/exp_threads 100002 1000; VMD_SDF=1 /tmp/exp_threads 100002 10000; /tmp/exp_threads 1000002 200
echo "== read-ahead off"
VMD_OPTS="readahead=0" /tmp/exp_threads 100002 1000; VMD_OPTS="readahead=0" VMD_SDF=1 /tmp/exp_threads 100002 10000
echo "== first region 256 / growth 4"
VMD_OPTS="readahead_frames=256" /tmp/exp_threads 100002 1000; VMD_OPTS="readahead_frames=256" VMD_SDF=1 /tmp/exp_threads 100002 10000
echo "== first region 64 / growth 8"
VMD_OPTS="readahead_frames=64 readahead_growth=8" /tmp/exp_threads 100002 1000; VMD_OPTS="readahead_frames=64 readahead_growth=8" VMD_SDF=1 /tmp/exp_threads 100002 10000
echo "== blocks of 64 (rdf) / 512 (sdf)"
VMD_OPTS="readahead_block=64" /tmp/exp_threads 100002 1000; VMD_OPTS="readahead_block=512" VMD_SDF=1 /tmp/exp_threads 100002 10000
} 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "read_ahead or pool_threads or filtered or concurrently or sharding" > $O/pytest_ra.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ra.log
