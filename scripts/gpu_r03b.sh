#!/bin/bash
# Round 3, second GPU call: k_xtc_wave with several waves per frame, the asynchronous staging pipeline, the compressed-resident
# trajectory; c2 end to end from an XTC file.
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r03b.sh [tag]'
TAG=${1:-r03b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > $OUT/device.txt

echo "== pytest -m gpu (xdr)"
timeout 600 python -m pytest tests/test_xdr.py tests/test_zzz_xdr_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_xdr.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest_gpu_xdr.log

echo "== decoders in isolation"
timeout 600 python scripts/exp_xtc_decode.py $OUT/xtc_decode_isolated.txt --quick > $OUT/exp.log 2>&1; echo "exp rc=$?"
tail -3 $OUT/exp.log

run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 "$@" > $OUT/bench_c2_$n.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$n.json'));print('$n', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), {k: round(v, 1) for k, v in d.get('kernel_ms', {}).items()})"
}
echo "== c2 end to end"
run xtc_host32 --traj xtc --opt load_threads=32
run xtc_dev3 --traj xtc --opt xtc_device_decode=3
run xtc_dev3_t32 --traj xtc --opt xtc_device_decode=3 --opt load_threads=32
run xtc_dev3_t32_s256 --traj xtc --opt xtc_device_decode=3 --opt load_threads=32 --opt stage_frames=256
run xtc_dev3_t16_s64 --traj xtc --opt xtc_device_decode=3 --opt load_threads=16 --opt stage_frames=64
run xtc_resident --traj xtc-resident
run xtc_resident_s256 --traj xtc-resident --opt stage_frames=256
run xtc_resident_s500 --traj xtc-resident --opt stage_frames=500
run pinned --traj pinned
run device --traj device
tail -3 $OUT/bench_xtc.err

echo "== rocprofv3 kernel stats of the resident run"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_xtc -o xtc -- python $R/bench.py --workload c2 --traj xtc-resident --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_xtc.log 2>&1; echo "rocprof rc=$?"
for f in $(find $OUT/prof_xtc -name "*kernel_stats.csv"); do head -8 $f; done
find $OUT/prof_xtc -name "*kernel_trace.csv" -size +5M -delete
echo done
