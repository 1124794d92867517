#!/bin/bash
# r03s: FIRST pass over an XTC file (no checkpoints): walks of several batches side by side on their own streams
T=${1:-r03s}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_zzz_xdr_gpu.py tests/test_xdr.py -m gpu -x -q > $O/pytest_xdr.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_xdr.log
run() {  tag=$1; shift
  timeout 600 python bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']; fp=d['config'].get('first_pass')
print('$tag', round(d['value']), 'frames/s; first step', round(fp['frames_per_s']) if fp else None, {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
}
run default
run cold_streams0 --opt xtc_cold_streams=0
run nock_streams1 --opt xtc_checkpoints=0
run nock_streams0 --opt xtc_checkpoints=0 --opt xtc_cold_streams=0
run nock_streams1_s64 --opt xtc_checkpoints=0 --opt stage_frames=64
run nock_streams1_s256 --opt xtc_checkpoints=0 --opt stage_frames=256
run rw_nock_streams1 --opt xtc_checkpoints=0 --rigid-water
run rw_nock_streams0 --opt xtc_checkpoints=0 --opt xtc_cold_streams=0 --rigid-water
tail -3 $O/err.log
