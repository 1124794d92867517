#!/bin/bash
# r03q: the file-backed XTC path after the mapped-file DMA: first pass / re-evaluation, synthetic box and rigid water, against the
# pinned-block copy and the host-thread decode; kernel stats of the default configuration
T=${1:-r03q}; O=$PWD/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
run() {  # tag, args...
  tag=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']; fp=d['config'].get('first_pass')
print('$tag', round(d['value']), 'frames/s; first pass', round(fp['frames_per_s']) if fp else None, {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
}
run file_mapped --traj xtc
run file_copy --traj xtc --opt xtc_mapped=0
run file_host32 --traj xtc --opt xtc_device_decode=0 --opt load_threads=32
run resident --traj xtc-resident
run rw_file_mapped --traj xtc --rigid-water
run rw_file_copy --traj xtc --rigid-water --opt xtc_mapped=0
run rw_resident --traj xtc-resident --rigid-water
run floats_resident
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_xtc -o xtc -- python $R/bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 5 --warmup 2 > $O/prof_xtc.log 2>&1
for f in $(find $O/prof_xtc -name "*kernel_stats.csv"); do head -8 $f; done
find $O/prof_xtc -name "*kernel_trace.csv" -size +5M -delete
tail -3 $O/err.log
