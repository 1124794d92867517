#!/bin/bash
# round 5: the deferred-settle mode under the randomised programs on the MI355X
TAG=r05x; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
{
echo "Deferred-settle mode (option readahead_lone) on the MI355X, final tree (scripts/gpu_r05x.sh):"
timeout 900 python scripts/fuzz_pool.py 73000 73900 gpu lone 2>&1 | tail -2
g++ -std=c++17 -O2 tests/native/stress_readahead.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/stress_ra
for seed in 21 22 23 24; do timeout 600 /tmp/stress_ra 600 240 30000 $seed 2>&1 | tail -1; done
timeout 600 /tmp/stress_ra 200 240 30000 25 sdf 2>&1 | tail -1
} > $OUT/campaign.txt 2>&1
cat $OUT/campaign.txt | cut -c1-300
