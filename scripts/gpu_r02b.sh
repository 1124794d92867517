#!/bin/bash
# Round 2, second GPU call: parity tests (incl. full-size), default bench, A/B of this round's changes (two-level cell build,
# class decomposition of co-evaluated RDFs, SDF scatter addressing), kernel stats + PMC traffic of c3 / c2 with the new build.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r02b.sh [tag]'
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== VALU calibration (extra rows)"
[ -x build/valu_calib ] || hipcc --offload-arch=gfx950 -O3 -o build/valu_calib scripts/valu_calib.hip
timeout 300 ./build/valu_calib > $OUT/valu_calibration.txt 2>&1; echo "calib rc=$?"; tail -12 $OUT/valu_calibration.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
ab() {   # name, bench args...
  local name=$1; shift
  timeout 600 python bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/ab_$name.json 2>> $OUT/ab.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_$name.json"))
    print("%-28s %12.0f frames/s %9.3f ms/step  kernels %s" % ("$name", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["kernel_ms"].items()}))
except Exception as ex:
    print("$name FAILED", ex)
PY
}
echo "== A/B"
ab c3_new        --workload c3 --steps 6
ab c3_oldcells   --workload c3 --steps 6 --opt cells_pencil=0
ab c3_nxf_lds    --workload c3 --steps 6 --opt nxf_divisor=5
ab c2_new        --workload c2 --steps 20
ab c2_oldcells   --workload c2 --steps 20 --opt cells_pencil=0
ab c5_new        --workload c5 --steps 3
ab c5_noclasses  --workload c5 --steps 3 --opt rdf_classes=0
ab c5_old        --workload c5 --steps 3 --opt rdf_classes=0 --opt cells_pencil=0
ab c4_new        --workload c4 --steps 5
ab c4_ilp8       --workload c4 --steps 5 --opt sdf_ilp=8
ab c4_noarith    --workload c4 --steps 5 --opt sdf_arith=0
ab c4_noarith8   --workload c4 --steps 5 --opt sdf_arith=0 --opt sdf_ilp=8
ab c4_batch1024  --workload c4 --steps 5 --batch 1024
ab c4_dense      --workload c4 --steps 5 --opt sdf_dense=1
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
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -o c4 -- python $R/bench.py --workload c4 --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $OUT/prof_c4.log 2>&1
for f in $(find $OUT/prof_c4 -name "*kernel_stats.csv"); do cat $f; done
find $OUT/prof_c4 -name "*kernel_trace.csv" -size +5M -delete
cd $R
echo "== PMC c3 / c2 / c4 (new cell build)"
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_c3 --no-secondary > $OUT/pmc_c3.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_c3 c3 500 $OUT/pmc_traffic.json > /dev/null
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_c2 --workload c2 > $OUT/pmc_c2.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_c2 c2 500 $OUT/pmc_traffic.json > /dev/null
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_c4 --workload c4 > $OUT/pmc_c4.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_c4 c4 500 $OUT/pmc_traffic.json > /dev/null
python - <<PY
import json
t = json.load(open("$OUT/pmc_traffic.json"))
for w in t:
    print(w, {k: round(v["hbm_bytes_per_launch_read_x2"] / t[w]["frames_per_launch"] / 1e6, 3) for k, v in t[w]["kernels"].items() if k.startswith("k_")})
PY
echo done
