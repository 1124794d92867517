#!/bin/bash
T=${1:-r03t2}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  tag=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']; fp=d['config'].get('first_pass')
print('$tag', round(d['value']), 'frames/s; first step', round(fp['frames_per_s']) if fp else None, {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
}
run file_rec1 --traj xtc
run file_rec0 --traj xtc --opt xtc_records=0
run rw_file_rec1 --traj xtc --rigid-water
run rw_file_rec0 --traj xtc --rigid-water --opt xtc_records=0
run resident_rec1 --traj xtc-resident
run resident_rec0 --traj xtc-resident --opt xtc_records=0
tail -3 $O/err.log
