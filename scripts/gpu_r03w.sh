#!/bin/bash
# r03w: HEAD after the container was re-created: XTC device-decode tests + the file / resident lines of r03v (consecutive sections per wave)
T=${1:-r03w}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_zzz_xdr_gpu.py tests/test_xdr.py -m gpu -x -q > $O/pytest_xdr.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_xdr.log
run() {  tag=$1; wl=$2; shift; shift
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 5 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']; fp=d['config'].get('first_pass')
print('$tag', round(d['value']), 'frames/s; first step', round(fp['frames_per_s']) if fp else None, {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
}
run c2_xtc c2 --traj xtc
run c2_resident c2 --traj xtc-resident
run c2_rw_xtc c2 --traj xtc --rigid-water
run c3_xtc c3 --frames 200 --traj xtc
tail -5 $O/err.log
for S in 125 50; do timeout 300 python scripts/exp_filtered.py c2 $S > $O/filtered_c2_$S.json 2>> $O/err.log; cat $O/filtered_c2_$S.json; done
timeout 300 python scripts/exp_filtered.py c3 125 > $O/filtered_c3_125.json 2>> $O/err.log; cat $O/filtered_c3_125.json
