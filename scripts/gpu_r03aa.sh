#!/bin/bash
# r03aa: 32-bit path for small XTC triples + reciprocal division for wide numbers over large divisors: XTC device tests, then the
# decode time in the file / compressed-resident lines of c2 and c3
T=${1:-r03aa}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_zzz_xdr_gpu.py tests/test_xdr.py -m gpu -x -q > $O/pytest_xdr.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_xdr.log
run() {  tag=$1; wl=$2; shift; shift
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 5 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY | tee -a $O/lines.txt
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']; fp=d['config'].get('first_pass')
print('$tag', round(d['value']), 'frames/s; first step', round(fp['frames_per_s']) if fp else None, {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
}
run c2_xtc c2 --traj xtc
run c2_resident c2 --traj xtc-resident
run c2_rw_xtc c2 --traj xtc --rigid-water
run c2_rw_resident c2 --traj xtc-resident --rigid-water
run c3_xtc c3 --frames 200 --traj xtc
run c3_resident c3 --frames 200 --traj xtc-resident
grep -v amdgpu.ids $O/err.log | tail -5
