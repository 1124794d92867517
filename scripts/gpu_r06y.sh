#!/bin/bash
# round 6, last check on the final host code (kernels unchanged since scripts/gpu_r06z.sh: the committed counters stay valid): GPU suite, smoke, default line
T=${1:-r06y}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step frac", round(d["roofline"]["frac"], 4), "cpu", round(d["cpu_baseline"]["value"], 1), [round(x, 1) for x in d["cpu_baseline"]["samples"]], "gpu/cpu", round(d["gpu_over_cpu"], 1), "counters current", d["roofline"]["traffic_counters_match_kernel_source"], d["cpu_baseline"]["sample"][:60])
for k, v in d.get("secondary", {}).items():
    if k == "c1": print("c1", v["gpu_ms"]["pool_threads_16_grain_1"], v["gpu_ms"]["one_call"], v["cpu_ms"]["pool"], v["gpu_over_cpu"]); continue
    print(k, round(v["value"]), round(v["ms_per_step"], 3), v.get("strong_scaling_bound_8_gpus"), v.get("rank_part_ms"))
PY
