#!/bin/bash
# Round 3: c4 step A/B - float view written straight into pinned host pages (sdf_direct_view), non-temporal gathers (sdf_nt).
TAG=${1:-r03i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c4 --no-cpu-baseline --steps 10 --warmup 3 "$@" > $OUT/bench_c4_$n.json 2>> $OUT/bench.err
  python -c "import json;d=json.load(open('$OUT/bench_c4_$n.json'));print('$n', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms/step', {k: round(v/d['steps'], 3) for k, v in d.get('kernel_ms', {}).items()}, 'kernel frac', round(d['roofline']['frac'],4), 'step frac', round(d['roofline']['step_level']['frac'],4))"
}
for rep in 1 2; do
  run default_$rep
  run copy_view_$rep --opt sdf_direct_view=0
  run nt_$rep --opt sdf_nt=1
  run nt_ilp8_$rep --opt sdf_nt=1 --opt sdf_ilp=8
done
timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "sdf or volume or c4" > $OUT/pytest_sdf.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_sdf.log
echo done
