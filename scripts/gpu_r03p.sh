#!/bin/bash
# r03p: timeline of the file-backed XTC path (kernel + memory-copy trace, no counters)
T=${1:-r03p}; O=$PWD/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o xtc -- python $R/bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 3 --warmup 2 > $O/trace.log 2>&1
echo "rc=$?"; tail -2 $O/trace.log
find $O/trace -name "*.csv" | xargs ls -la
