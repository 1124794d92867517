#!/bin/bash
# r03ah: decoder checkpoints in a sidecar file (vmd_ckcache_save / _load): GPU test, then the first pass of a process over an XTC file
# without and with a sidecar written by an earlier pass (c2, rigid water, c3)
T=${1:-r03ah}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_xdr.py tests/test_zzz_xdr_gpu.py tests/test_abi.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() {  tag=$1; wl=$2; shift; shift
  timeout 900 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 3 --warmup 2 "$@" > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY | tee -a $O/lines.txt
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
fp=d['config'].get('first_pass')
print('%-22s first pass of the process %7d frames/s (%.1f ms)   steady state %7d frames/s' % ('$tag', round(fp['frames_per_s']), fp['ms'], round(d['value'])))
PY
}
run c2_xtc c2 --traj xtc
run c2_xtc_sidecar c2 --traj xtc --ck-sidecar
run c2_rw_xtc c2 --traj xtc --rigid-water
run c2_rw_xtc_sidecar c2 --traj xtc --rigid-water --ck-sidecar
run c3_xtc c3 --frames 200 --traj xtc
run c3_xtc_sidecar c3 --frames 200 --traj xtc --ck-sidecar
grep -v amdgpu.ids $O/err.log | tail -5
