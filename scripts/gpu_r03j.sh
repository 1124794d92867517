#!/bin/bash
# Round 3: k_xtc_wave at wave priority 3 under a pair grid that leaves it a slot (rdf_blocks_decode = 1536 / 1280) against the default.
TAG=${1:-r03j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 "$@" > $OUT/bench_c2_$n.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$n.json'));print('$n', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), {k: round(v, 1) for k, v in d.get('kernel_ms', {}).items()})"
}
for rw in "" "--rigid-water"; do
  echo "== c2 $rw"
  for blk in 0 1536 1280; do
    for S in 32 64; do
      run resident_s${S}_b$blk$rw --traj xtc-resident $rw --opt rdf_blocks_decode=$blk --opt stage_frames=$S
    done
    run file_b$blk$rw --traj xtc $rw --opt rdf_blocks_decode=$blk
    run file_s32_b$blk$rw --traj xtc $rw --opt rdf_blocks_decode=$blk --opt stage_frames=32
  done
  run resident$rw --traj xtc-resident $rw
done
timeout 300 python -m pytest tests/test_xdr.py tests/test_zzz_xdr_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
echo done
