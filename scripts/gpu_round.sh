#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench (variants), rocprofv3 kernel stats.  Outputs under gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh [tag]'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
nproc >> $OUT/device.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core" >> $OUT/device.txt

echo "== pytest -m gpu"; 
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
echo "== bench c2 (default run, with cpu baseline)"
timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"; cat $OUT/bench_c2.json; tail -3 $OUT/bench_c2.err
for v in 0 1; do
  echo "== bench c2 variant $v"
  timeout 300 python bench.py --variant $v --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c2_v$v.json 2>> $OUT/bench_c2.err; cat $OUT/bench_c2_v$v.json
done
for b in 32 64 128; do
  echo "== bench c2 batch $b"
  timeout 300 python bench.py --batch $b --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_c2_b$b.json 2>> $OUT/bench_c2.err; cat $OUT/bench_c2_b$b.json
done
echo "== bench c3 (200 frames)"
timeout 600 python bench.py --workload c3 --frames 200 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; cat $OUT/bench_c3.json; tail -3 $OUT/bench_c3.err
echo "== rocprofv3 kernel stats (c2)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o c2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_c2.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof_c2 -name "*stats*" | head; for f in $(find $OUT/prof_c2 -name "*kernel_stats.csv"); do head -12 $f; done
# keep only the small summaries (traces can be large)
find $OUT/prof_c2 -name "*kernel_trace.csv" -size +20M -delete
