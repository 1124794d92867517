#!/bin/bash
# Full round evidence in one gpurun call: GPU parity tests, smoke, default bench (with CPU baseline), the other workloads,
# rocprofv3 kernel stats of the default bench command, PMC passes.  Outputs under gpurun_out/<tag>/.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh [tag]'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > $OUT/device.txt

echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== bench (default = c2, with cpu baseline)"
timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"; cat $OUT/bench_c2.json
for w in c3 c4 c5; do
  echo "== bench $w"
  timeout 900 python bench.py --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; cat $OUT/bench_$w.json
done
echo "== bench c2, trajectory in pinned host memory (PCIe-inclusive)"
timeout 300 python bench.py --traj pinned --no-cpu-baseline > $OUT/bench_c2_pinned.json 2>> $OUT/bench_c2.err; cat $OUT/bench_c2_pinned.json
echo "== bench c2 from trajectory files (file -> host decode threads -> pinned staging -> PCIe): dcd, trr, xtc; xtc with 8 / 32 / 64 decode threads"
for t in dcd trr xtc; do
  timeout 600 python bench.py --traj $t --no-cpu-baseline --steps 5 > $OUT/bench_c2_$t.json 2>> $OUT/bench_c2.err; cat $OUT/bench_c2_$t.json
done
for n in 8 32 64; do
  timeout 600 python bench.py --traj xtc --no-cpu-baseline --steps 5 --opt load_threads=$n > $OUT/bench_c2_xtc_t$n.json 2>> $OUT/bench_c2.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_xtc_t$n.json'));print('xtc load_threads=$n', round(d['value']), 'frames/s')"
done
echo "== bench c2 from an XTC file, frames decompressed on the device (k_xtc_decode): compressed bytes cross PCIe"
timeout 600 python bench.py --traj xtc --no-cpu-baseline --steps 5 --opt xtc_device_decode=1 > $OUT/bench_c2_xtc_device.json 2>> $OUT/bench_c2.err; cat $OUT/bench_c2_xtc_device.json
for ch in 128 256 1024; do
  timeout 600 python bench.py --traj xtc --no-cpu-baseline --steps 5 --opt xtc_device_decode=2 --opt xtc_chunk=$ch > $OUT/bench_c2_xtc_device2_$ch.json 2>> $OUT/bench_c2.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_xtc_device2_$ch.json'));print('xtc two-pass device decode, chunk $ch:', round(d['value']), 'frames/s')"
done
echo "== rocprofv3 --kernel-trace --stats of the default bench command"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -o c2 -- python $R/bench.py --no-cpu-baseline > $OUT/prof_c2.log 2>&1; echo "rocprof rc=$?"
for f in $(find $OUT/prof_c2 -name "*kernel_stats.csv"); do cat $f; done
tail -1 $OUT/prof_c2.log
find $OUT/prof_c2 -name "*kernel_trace.csv" -size +20M -delete
echo "== PMC"
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc > $OUT/pmc.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc c2 500 $OUT/pmc_traffic.json
echo "== PMC (c4: SDF scatter)"
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_c4 --workload c4 > $OUT/pmc_c4.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_c4 c4 500 $OUT/pmc_traffic.json | tail -3
