#!/bin/bash
# Round 3: decoder checkpoints (first pass leaves them, later passes decode every frame in sections) - in isolation and end to end.
TAG=${1:-r03k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_xdr.py tests/test_zzz_xdr_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
echo "== decoders in isolation"
timeout 600 python scripts/exp_xtc_decode.py $OUT/xtc_decode_isolated.txt --quick > $OUT/exp.log 2>&1; echo "exp rc=$?"
grep -E "^==|checkpoints|, +(1|0) waves" $OUT/xtc_decode_isolated.txt | cut -c1-210
run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 "$@" > $OUT/bench_c2_$n.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$n.json'));print('$n', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), {k: round(v, 1) for k, v in d.get('kernel_ms', {}).items()})"
}
for rw in "" "--rigid-water"; do
  echo "== c2 $rw"
  run file$rw --traj xtc $rw
  run file_nock$rw --traj xtc $rw --opt xtc_checkpoints=0
  run file_s64$rw --traj xtc $rw --opt stage_frames=64
  run file_s256$rw --traj xtc $rw --opt stage_frames=256
  run resident$rw --traj xtc-resident $rw
  run resident_nock$rw --traj xtc-resident $rw --opt xtc_checkpoints=0
  run resident_s64$rw --traj xtc-resident $rw --opt stage_frames=64
  run resident_s256$rw --traj xtc-resident $rw --opt stage_frames=256
  run resident_s256_b0$rw --traj xtc-resident $rw --opt stage_frames=256 --opt rdf_blocks_decode=0
done
run host32 --traj xtc --opt xtc_device_decode=0 --opt load_threads=32
run pinned --traj pinned
echo done
