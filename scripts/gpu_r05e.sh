#!/bin/bash
# round 5, call e: deferred settle (option readahead_lone) on the MI355X - parity cases, the C++ stress, the call-pattern table with the new column
TAG=r05e; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_native.py -m gpu -q -x -p no:cacheprovider -k "read_ahead or readahead or pool" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads && { /tmp/exp_threads 100002 1000; VMD_SDF=1 /tmp/exp_threads 100002 10000; } > $OUT/readahead_call_pattern.txt 2>&1; cat $OUT/readahead_call_pattern.txt | cut -c1-1200
