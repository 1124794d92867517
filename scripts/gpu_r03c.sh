#!/bin/bash
# Round 3, third GPU call: k_xtc_wave with the two-bank register window (no stall per block), host I/O experiment,
# c2 end to end from an XTC file / compressed-resident.
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > $OUT/device.txt

echo "== pytest -m gpu (xdr)"
timeout 600 python -m pytest tests/test_xdr.py tests/test_zzz_xdr_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_xdr.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest_gpu_xdr.log

echo "== host I/O"
[ -x build/exp_hostio ] || hipcc -O2 -o build/exp_hostio scripts/exp_hostio.cpp -lpthread
timeout 300 ./build/exp_hostio 512 > $OUT/hostio.txt 2>&1; echo "hostio rc=$?"; cat $OUT/hostio.txt

echo "== decoders in isolation"
timeout 600 python scripts/exp_xtc_decode.py $OUT/xtc_decode_isolated.txt --quick > $OUT/exp.log 2>&1; echo "exp rc=$?"
grep -E "^==|waves per frame" $OUT/xtc_decode_isolated.txt | awk '{print}' | cut -c1-150
tail -2 $OUT/exp.log

run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 "$@" > $OUT/bench_c2_$n.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$n.json'));print('$n', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), {k: round(v, 1) for k, v in d.get('kernel_ms', {}).items()})"
}
echo "== c2 end to end"
run xtc_host32 --traj xtc --opt load_threads=32
run xtc_dev3_t32_s128 --traj xtc --opt xtc_device_decode=3 --opt load_threads=32
run xtc_dev3_t32_s256 --traj xtc --opt xtc_device_decode=3 --opt load_threads=32 --opt stage_frames=256
run xtc_dev3_t16_s256 --traj xtc --opt xtc_device_decode=3 --opt load_threads=16 --opt stage_frames=256
run xtc_resident_s128 --traj xtc-resident
run xtc_resident_s256 --traj xtc-resident --opt stage_frames=256
run xtc_resident_s500 --traj xtc-resident --opt stage_frames=500
run xtc_resident_s1000 --traj xtc-resident --opt stage_frames=1000
tail -3 $OUT/bench_xtc.err
echo done
