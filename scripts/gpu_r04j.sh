#!/bin/bash
# closing check of round 4: the whole -m gpu suite, smoke, the driver-style default line (counters of profiles/pmc_traffic.json must match the kernel source)
T=${1:-r04j}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], "counters match", d["roofline"]["traffic_counters_match_kernel_source"], "cpu", round(d["cpu_baseline"]["value"], 1), "gpu/cpu", round(d["gpu_over_cpu"], 1))
print("cell build", d["cell_build"])
for k, v in d.get("secondary", {}).items(): print(k, round(v["value"]), round(v["ms_per_step"], 3), "kernel frac", round(v["roofline"]["frac"], 4), "step frac", round(v["roofline"]["step_level"]["frac"], 4))
PY
