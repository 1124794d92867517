#!/bin/bash
# rocprofv3 kernel stats of one bench command.  usage: gpurun -- 'bash scripts/gpu_kstats.sh <tag> <bench args...>'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $R/bench.py --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
echo "rc=$?"
for f in $(find $OUT -name "*kernel_stats.csv"); do cut -d, -f1-4 $f; done
find $OUT -name "*kernel_trace.csv" -size +5M -delete
tail -1 $OUT/log.txt | cut -c1-300
