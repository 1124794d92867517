#!/bin/bash
T=${1:-r03u3}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  tag=$1; wl=$2; shift; shift
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 4 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']; fp=d['config'].get('first_pass')
print('$tag', round(d['value']), 'frames/s; first step', round(fp['frames_per_s']) if fp else None, {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
}
run c3_xtc c3 --frames 200 --traj xtc
run c3_xtc_cold c3 --frames 200 --traj xtc --opt xtc_checkpoints=0
run c2_xtc_cold c2 --traj xtc --opt xtc_checkpoints=0
run c2_xtc c2 --traj xtc
tail -5 $O/err.log
