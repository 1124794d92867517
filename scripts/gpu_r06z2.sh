#!/bin/bash
# round 6, supplement to the evidence call after the capacity floor (host code only; vmd_kernels.hip is byte for byte what r06z profiled): c5's cell
# builds per step, its counters and kernel stats again, the GPU suite, and the default line on the final tree
T=${1:-r06z2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python scripts/exp_c5_builds.py 1000 > $O/c5_builds.txt 2>&1; tail -6 $O/c5_builds.txt
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
bash $R/scripts/gpu_pmc.sh ${T}_pmc_c5 --workload c5 --no-secondary > $O/pmc_c5.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${T}_pmc_c5 c5 1000 $O/pmc_traffic.json 2 1000 > /dev/null
cp $R/gpurun_out/${T}_pmc_c5/summary.txt $O/pmc_summary_c5.txt 2>/dev/null; rm -rf $R/gpurun_out/${T}_pmc_c5
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python $R/bench.py --workload c5 --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $O/prof_c5.log 2>&1
f=$(find $O/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_c5.csv; rm -rf $O/prof_c5
cd $R
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step frac", round(d["roofline"]["frac"], 4), "cpu", round(d["cpu_baseline"]["value"], 1), [round(x, 1) for x in d["cpu_baseline"]["samples"]], "gpu/cpu", round(d["gpu_over_cpu"], 1), "counters current", d["roofline"]["traffic_counters_match_kernel_source"])
for k, v in d.get("secondary", {}).items():
    if k == "c1": print("c1", v["gpu_ms"]["pool_threads_16_grain_1"], v["gpu_ms"]["one_call"], v["cpu_ms"]["pool"], v["gpu_over_cpu"]); continue
    print(k, round(v["value"]), round(v["ms_per_step"], 3), v.get("strong_scaling_bound_8_gpus"), v.get("rank_part_ms"), (v.get("cell_build") or {}).get("launches_per_step"), (v.get("cell_build") or {}).get("traffic_ratio"), (v.get("cell_build") or {}).get("bucket_overflows_since_creation"))
PY
du -sh $O
