#!/bin/bash
# Round 3, evidence call: all GPU tests, smoke, the default bench line as the driver runs it, rocprofv3 kernel stats of the
# same command, PMC traffic of c3 / c2 / c4, strong-scaling code path at N = 1.
TAG=${1:-r03zz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > $OUT/device.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== bench default (c3 + secondary + cpu baseline), as the driver runs it"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step", d["kernel_ms"], "cpu", d.get("cpu_baseline", {}).get("value"), "valu", d["roofline"]["valu"] and round(d["roofline"]["valu"]["frac"], 3))
for k, v in d.get("secondary", {}).items():
    print(k, round(v["value"]), "frames/s", round(v["ms_per_step"], 3), "ms/step", v["kernel_ms"], "kernel frac", round(v["roofline"]["frac"], 4), "step frac", round(v["roofline"]["step_level"]["frac"], 4))
PY
tail -3 $OUT/bench_default.err
echo "== strong-scaling path at N = 1 (c4, 10 000 frames as one shard) and c4 with 5 steps"
timeout 600 python bench.py --workload c4 --scaling strong --steps 5 --no-cpu-baseline > $OUT/bench_c4_strong.json 2>> $OUT/bench_default.err
python -c "import json;d=json.load(open('$OUT/bench_c4_strong.json'));print('c4 strong', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms/step', d['kernel_ms'], 'step frac', round(d['roofline']['step_level']['frac'],4))"
echo "== rocprofv3 --kernel-trace --stats of the default workload"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -o c3 -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $OUT/prof_c3.log 2>&1; echo "rocprof rc=$?"
for f in $(find $OUT/prof_c3 -name "*kernel_stats.csv"); do cat $f; done
find $OUT/prof_c3 -name "*kernel_trace.csv" -size +5M -delete
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -o c4 -- python $R/bench.py --workload c4 --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $OUT/prof_c4.log 2>&1
for f in $(find $OUT/prof_c4 -name "*kernel_stats.csv"); do grep -v rocclr_copyBuffer $f | head -8; done
find $OUT/prof_c4 -name "*kernel_trace.csv" -size +5M -delete
cd $R
echo "== c2 from an XTC file / compressed-resident (device decode, checkpoints) + kernel stats of the file path"
for t in xtc xtc-resident; do
  timeout 600 python bench.py --workload c2 --traj $t --no-cpu-baseline --steps 10 --warmup 2 > $OUT/bench_c2_$t.json 2>> $OUT/bench_default.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$t.json'));print('c2 $t', round(d['value']), 'frames/s', round(d['ms_per_step'],2), 'ms/step', {k: round(v/d['steps'], 2) for k, v in d['kernel_ms'].items()}, 'first pass', round(d['config'].get('first_pass', {}).get('frames_per_s', 0)))"
done
timeout 600 python bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 5 --opt xtc_device_decode=0 --opt load_threads=32 > $OUT/bench_c2_xtc_host32.json 2>> $OUT/bench_default.err
python -c "import json;d=json.load(open('$OUT/bench_c2_xtc_host32.json'));print('c2 xtc host threads', round(d['value']), 'frames/s')"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_xtc -o xtc -- python $R/bench.py --workload c2 --traj xtc --no-cpu-baseline --steps 5 --warmup 2 > $OUT/prof_xtc.log 2>&1
for f in $(find $OUT/prof_xtc -name "*kernel_stats.csv"); do grep -v rocclr_fill $f | head -6; done
find $OUT/prof_xtc -name "*kernel_trace.csv" -size +5M -delete
cd $R
echo "== PMC c3 / c2 / c4"
for w in c3 c2 c4; do
  extra="--workload $w"; [ $w = c3 ] && extra="--no-secondary"
  bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_$w $extra > $OUT/pmc_$w.log 2>&1
  python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_$w $w 500 $OUT/pmc_traffic.json > /dev/null
done
python - <<PY
import json
t = json.load(open("$OUT/pmc_traffic.json"))
for w in t:
    print(w, {k: round(v["hbm_bytes_per_launch_read_x2"] / t[w]["frames_per_launch"] / 1e6, 3) for k, v in t[w]["kernels"].items() if k.startswith("k_")})
PY
echo done
