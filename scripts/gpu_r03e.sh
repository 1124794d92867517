#!/bin/bash
# Round 3, fifth GPU call: decode UNDER the pair kernel (80-VGPR k_xtc_wave, pair grid of 6 blocks per CU while decoding, ramped batches).
TAG=${1:-r03e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu (xdr, native shim)"
timeout 600 python -m pytest tests/test_xdr.py tests/test_zzz_xdr_gpu.py tests/test_native.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_xdr.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest_gpu_xdr.log
echo "== decoders in isolation"
timeout 600 python scripts/exp_xtc_decode.py $OUT/xtc_decode_isolated.txt --quick > $OUT/exp.log 2>&1; echo "exp rc=$?"
grep -E "^==|, +(1|4|0) waves per frame" $OUT/xtc_decode_isolated.txt | cut -c1-120
run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 "$@" > $OUT/bench_c2_$n.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$n.json'));print('$n', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), {k: round(v, 1) for k, v in d.get('kernel_ms', {}).items()})"
}
for rw in "" "--rigid-water"; do
  echo "== c2 end to end $rw"
  run xtc_host32$rw --traj xtc $rw --opt load_threads=32
  for blk in 2048 1536 1280; do
    run xtc_dev3_b$blk$rw --traj xtc $rw --opt xtc_device_decode=3 --opt load_threads=16 --opt rdf_blocks_decode=$blk
    run xtc_resident_b$blk$rw --traj xtc-resident $rw --opt rdf_blocks_decode=$blk
  done
  run xtc_dev3_s64$rw --traj xtc $rw --opt xtc_device_decode=3 --opt load_threads=16 --opt stage_frames=64
  run xtc_resident_s64$rw --traj xtc-resident $rw --opt stage_frames=64
  run xtc_resident_s256$rw --traj xtc-resident $rw --opt stage_frames=256
done
tail -3 $OUT/bench_xtc.err
echo done
