#!/bin/bash
T=${1:-r03o}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "xtc_ramp=0" "xtc_ramp=1" "xtc_ramp=1 rdf_blocks_decode=0" "xtc_ramp=1 stage_frames=96" "xtc_ramp=1 stage_frames=192" "xtc_ramp=1 --rigid-water"; do
  tag=$(echo $v | tr '= ' '__' | tr -d '-')
  opt=""; tr="--traj xtc"; for w in $v; do case $w in --traj=*) tr="--traj ${w#--traj=}";; --*) opt="$opt $w";; *) opt="$opt --opt $w";; esac; done
  timeout 600 python bench.py --workload c2 $tr --no-cpu-baseline --steps 5 --warmup 2 $opt > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']
print('$tag', round(d['value']), 'frames/s', {a: round(b/s,2) for a,b in k.items()})
PY
done
tail -3 $O/err.log
