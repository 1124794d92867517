#!/bin/bash
# Round 3, sixth GPU call: batch plan of the device-decode paths (no ramp), pair grid 2048 vs 1536 blocks while decoding; kernel stats.
TAG=${1:-r03f}
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
  echo "== c2 end to end $rw"
  for blk in 2048 1536; do
    run xtc_dev3_b$blk$rw --traj xtc $rw --opt xtc_device_decode=3 --opt load_threads=16 --opt rdf_blocks_decode=$blk
    run xtc_dev3_s64_b$blk$rw --traj xtc $rw --opt xtc_device_decode=3 --opt load_threads=16 --opt rdf_blocks_decode=$blk --opt stage_frames=64
    run xtc_resident_s64_b$blk$rw --traj xtc-resident $rw --opt rdf_blocks_decode=$blk --opt stage_frames=64
  done
  run xtc_resident$rw --traj xtc-resident $rw
done
echo "== rocprofv3 kernel stats: c2 from the XTC file, device decode"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_xtc -o xtc -- python $R/bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 3 --warmup 1 --opt xtc_device_decode=3 > $OUT/prof_xtc.log 2>&1; echo "rocprof rc=$?"
for f in $(find $OUT/prof_xtc -name "*kernel_stats.csv"); do head -6 $f; done
find $OUT/prof_xtc -name "*kernel_trace.csv" -size +5M -delete
tail -3 $OUT/bench_xtc.err | grep -v amdgpu.ids
echo done
