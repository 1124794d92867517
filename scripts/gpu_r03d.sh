#!/bin/bash
# Round 3, fourth GPU call: the three-engine staging ring (PCIe / decode / pair) for file-backed XTC, synthetic and rigid water.
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu (xdr)"
timeout 600 python -m pytest tests/test_xdr.py tests/test_zzz_xdr_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_xdr.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest_gpu_xdr.log
run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 "$@" > $OUT/bench_c2_$n.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$n.json'));print('$n', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), {k: round(v, 1) for k, v in d.get('kernel_ms', {}).items()})"
}
echo "== c2 end to end, synthetic box"
run xtc_host32 --traj xtc --opt load_threads=32
run xtc_dev3_s128 --traj xtc --opt xtc_device_decode=3 --opt load_threads=16
run xtc_dev3_s256 --traj xtc --opt xtc_device_decode=3 --opt load_threads=16 --opt stage_frames=256
run xtc_dev3_s512 --traj xtc --opt xtc_device_decode=3 --opt load_threads=16 --opt stage_frames=512
run xtc_resident_s1000 --traj xtc-resident --opt stage_frames=1000
echo "== c2 end to end, rigid water"
run rw_xtc_host32 --traj xtc --rigid-water --opt load_threads=32
run rw_xtc_dev3_s128 --traj xtc --rigid-water --opt xtc_device_decode=3 --opt load_threads=16
run rw_xtc_dev3_s256 --traj xtc --rigid-water --opt xtc_device_decode=3 --opt load_threads=16 --opt stage_frames=256
run rw_xtc_dev3_s512 --traj xtc --rigid-water --opt xtc_device_decode=3 --opt load_threads=16 --opt stage_frames=512
run rw_xtc_resident_s256 --traj xtc-resident --rigid-water --opt stage_frames=256
run rw_xtc_resident_s1000 --traj xtc-resident --rigid-water --opt stage_frames=1000
tail -3 $OUT/bench_xtc.err
echo done
