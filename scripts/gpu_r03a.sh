#!/bin/bash
# Round 3, first GPU call: the wave-per-frame XTC decoder (k_xtc_wave) - fixtures on the GPU, decoders timed in isolation,
# c2 end to end from an XTC file (host threads vs device decode), kernel stats.
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r03a.sh [tag]'
TAG=${1:-r03a}
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
timeout 600 python scripts/exp_xtc_decode.py $OUT/xtc_decode_isolated.txt > $OUT/exp.log 2>&1; echo "exp rc=$?"
cat $OUT/xtc_decode_isolated.txt; tail -3 $OUT/exp.log

echo "== c2 end to end from an XTC file"
for o in "xtc_device_decode=3" "load_threads=32" "xtc_device_decode=3+load_threads=32"; do
  timeout 600 python bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 5 $(echo $o | sed 's/^/--opt /; s/+/ --opt /g') > $OUT/bench_c2_xtc_$o.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_xtc_$o.json'));print('xtc $o', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), d.get('kernel_ms'))"
done
tail -3 $OUT/bench_xtc.err

echo "== rocprofv3 kernel stats of the device-decode run"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_xtc -o xtc -- python $R/bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 3 --warmup 1 --opt xtc_device_decode=3 > $OUT/prof_xtc.log 2>&1; echo "rocprof rc=$?"
for f in $(find $OUT/prof_xtc -name "*kernel_stats.csv"); do head -12 $f; done
find $OUT/prof_xtc -name "*kernel_trace.csv" -size +5M -delete
echo done
