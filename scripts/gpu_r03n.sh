#!/bin/bash
# r03n: compressed XTC frames DMA'd straight out of the mapped file (no host copy) against the pinned-block copy
T=${1:-r03n}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_zzz_xdr_gpu.py tests/test_xdr.py -m gpu -x -q > $O/pytest_xdr.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_xdr.log
for v in "xtc_mapped=0" "xtc_mapped=1" "xtc_mapped=1 stage_frames=256" "xtc_mapped=1 stage_frames=64" "xtc_mapped=1 --rigid-water" "xtc_mapped=0 --rigid-water"; do
  tag=$(echo $v | tr '= ' '__' | tr -d '-')
  opt=""; for w in $v; do case $w in --*) opt="$opt $w";; *) opt="$opt --opt $w";; esac; done
  timeout 600 python bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 5 --warmup 2 $opt > $O/bench_xtc_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_xtc_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']
print('$tag', round(d['value']), 'frames/s; first pass', d.get('first_pass'), {a: round(b/s,2) for a,b in k.items()})
PY
done
tail -3 $O/err.log
