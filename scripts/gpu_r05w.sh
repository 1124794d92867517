#!/bin/bash
# round 5, closing randomised campaign on the final tree (product build, MI355X)
TAG=r05w; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
{
echo "Closing randomised campaign of round 5 on the final code (1 x MI355X; scripts/gpu_r05w.sh):"
timeout 600 python scripts/fuzz_pool.py 71000 71600 gpu 2>&1 | tail -2
timeout 900 python scripts/fuzz_gpu.py 70000 70450 20 2>&1 | tail -3
} > $OUT/campaign.txt 2>&1
cat $OUT/campaign.txt
