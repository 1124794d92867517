#!/bin/bash
# r03ad: deferred batch completion (the next batch is queued before the host waits for the current one): full GPU suite, then A/B on the
# file-backed lines (defer_sync 0 / 1, staged batch sizes) and on host-staged trajectories
T=${1:-r03ad}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
run() {  tag=$1; wl=$2; shift; shift
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 5 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY | tee -a $O/lines.txt
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']; fp=d['config'].get('first_pass')
print('%-26s' % '$tag', round(d['value']), 'frames/s; first step', round(fp['frames_per_s']) if fp else None, {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
}
run c2_xtc_nodefer c2 --traj xtc --opt defer_sync=0
run c2_xtc c2 --traj xtc
run c2_xtc_stage64 c2 --traj xtc --opt stage_frames=64
run c2_xtc_stage96 c2 --traj xtc --opt stage_frames=96
run c2_resident_nodefer c2 --traj xtc-resident --opt defer_sync=0
run c2_resident c2 --traj xtc-resident
run c2_rw_xtc c2 --traj xtc --rigid-water
run c3_xtc_nodefer c3 --frames 200 --traj xtc --opt defer_sync=0
run c3_xtc c3 --frames 200 --traj xtc
run c2_trr_nodefer c2 --traj trr --opt defer_sync=0
run c2_trr c2 --traj trr
run c2_pinned c2 --traj pinned
run c3_default c3 --steps 5
grep -v amdgpu.ids $O/err.log | tail -5
