#!/bin/bash
# rocprofv3 kernel stats of the secondary workloads (c3's are part of scripts/gpu_r05s.sh); usage: gpurun -- 'bash scripts/gpu_kstats.sh [tag]'
T=${1:-r05z}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${T}_kstats; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for w in c2 c4 c5 c3d; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $O/prof_$w.log 2>&1
  find $O/prof_$w -name "*kernel_trace.csv" -delete; find $O/prof_$w -name "*agent_info.csv" -delete
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$w.csv && echo "== $w" && head -6 $O/kernel_stats_$w.csv | cut -c1-160
done
