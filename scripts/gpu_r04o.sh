#!/bin/bash
# after the seed-8941 fix: the same seed range again and 1 200 new seeds, the whole -m gpu suite
T=${1:-r04o}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
for r in "8800 9200" "9200 9600" "9600 10000" "10000 10400"; do echo "# python scripts/fuzz_gpu.py $r 20"; timeout 1200 python scripts/fuzz_gpu.py $r 20 2>&1 | grep -v amdgpu.ids | tail -3; done
} | tee $O/campaign.txt
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-200
