#!/bin/bash
# r03m: where the eval thread's time goes on the file-backed XTC path (host timers next to the device event times)
T=${1:-r03m}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "default" "stage_frames=256" "load_threads=8" "load_threads=4"; do
  tag=$(echo $v | tr '=' '_')
  opt=""; [ "$v" != "default" ] && opt="--opt $v"
  timeout 600 python bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 5 --warmup 2 $opt > $O/bench_xtc_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_xtc_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']
print('$tag', round(d['value']), 'frames/s', {a: round(b/s,2) for a,b in k.items()})
PY
done
tail -3 $O/err.log
