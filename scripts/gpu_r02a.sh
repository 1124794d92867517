#!/bin/bash
# Round 2, first GPU call: VALU issue-rate calibration, GPU parity tests, the default bench line (c3 + secondary block + CPU
# baseline), rocprofv3 kernel stats + PMC passes of the default workload, XTC staging measurements.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r02a.sh [tag]'
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > $OUT/device.txt

echo "== VALU calibration"
[ -x build/valu_calib ] || hipcc --offload-arch=gfx950 -O3 -o build/valu_calib scripts/valu_calib.hip
timeout 300 ./build/valu_calib > $OUT/valu_calibration.txt 2>&1; echo "calib rc=$?"; cat $OUT/valu_calibration.txt

echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3

echo "== bench default (c3 + secondary + cpu baseline), as the driver runs it"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step", d["kernel_ms"], "cpu", d.get("cpu_baseline", {}).get("value"))
for k, v in d.get("secondary", {}).items():
    print(k, round(v["value"]), "frames/s", round(v["ms_per_step"], 3), "ms/step", v["kernel_ms"], "kernel frac", round(v["roofline"]["frac"], 4), "step frac", round(v["roofline"]["step_level"]["frac"], 4))
PY
tail -3 $OUT/bench_default.err

echo "== rocprofv3 --kernel-trace --stats of the default workload"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -o c3 -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $OUT/prof_c3.log 2>&1; echo "rocprof rc=$?"
for f in $(find $OUT/prof_c3 -name "*kernel_stats.csv"); do cat $f; done
find $OUT/prof_c3 -name "*kernel_trace.csv" -size +5M -delete
cd $R

echo "== PMC c3"
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_c3 --no-secondary > $OUT/pmc_c3.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_c3 c3 500 $OUT/pmc_traffic.json | tail -12
echo "== PMC c2"
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_c2 --workload c2 > $OUT/pmc_c2.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_c2 c2 500 $OUT/pmc_traffic.json | tail -12

echo "== XTC staging: c2 from an XTC file; host decode (8 / 32 threads), device decode one-pass / two-pass"
for o in "load_threads=8" "load_threads=32" "xtc_device_decode=1" "xtc_device_decode=2"; do
  timeout 600 python bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 5 --opt $o > $OUT/bench_c2_xtc_$o.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_xtc_$o.json'));print('xtc $o', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'))"
done
timeout 300 python bench.py --workload c2 --traj pinned --no-cpu-baseline --steps 10 > $OUT/bench_c2_pinned.json 2>> $OUT/bench_xtc.err
python -c "import json;d=json.load(open('$OUT/bench_c2_pinned.json'));print('pinned', round(d['value']), 'frames/s')"
echo done
