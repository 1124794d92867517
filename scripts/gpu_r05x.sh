#!/bin/bash
# round 5: the deferred-settle mode (and everything else the programs cover) under the randomised programs on the MI355X; usage: gpu_r05x.sh [tag] [seed0]
TAG=${1:-r05x}; S0=${2:-21}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
{
echo "Deferred-settle mode (option readahead_lone) on the MI355X, final tree (scripts/gpu_r05x.sh $TAG $S0):"
timeout 900 python scripts/fuzz_pool.py $((S0 * 1000)) $((S0 * 1000 + 1800)) gpu lone 2>&1 | tail -2
g++ -std=c++17 -O2 tests/native/stress_readahead.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/stress_ra
for k in 0 1 2 3 4 5 6 7; do timeout 600 /tmp/stress_ra 600 240 30000 $((S0 + k)) 2>&1 | tail -1; done
for k in 8 9; do timeout 600 /tmp/stress_ra 200 240 30000 $((S0 + k)) sdf 2>&1 | tail -1; done
} > $OUT/campaign.txt 2>&1
cat $OUT/campaign.txt | cut -c1-200
