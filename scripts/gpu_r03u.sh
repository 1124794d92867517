#!/bin/bash
# r03u: c3 (1 000 002 atoms, heavy-atom RDF) PCIe-inclusive: from an XTC file (compressed, mapped DMA, device decode) against floats in
# pinned host memory and the resident rate; 200 frames per step (a 1 GB file)
T=${1:-r03u}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  tag=$1; shift
  timeout 900 python bench.py --workload c3 --frames 200 --no-cpu-baseline --no-secondary --steps 4 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']; fp=d['config'].get('first_pass')
print('$tag', round(d['value']), 'frames/s; first step', round(fp['frames_per_s']) if fp else None, {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
}
run resident
run pinned --traj pinned
run xtc --traj xtc
run xtc_resident --traj xtc-resident
run xtc_host32 --traj xtc --opt xtc_device_decode=0 --opt load_threads=32
tail -5 $O/err.log
