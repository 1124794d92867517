#!/bin/bash
# the -m gpu suite + smoke + a short default bench line at the tree as it stands
T=${1:-suite}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step", "frac", round(d["roofline"]["frac"], 4), "counters match", d["roofline"].get("traffic_counters_match_kernel_source"), "cpu", round(d["cpu_baseline"]["value"], 1))
for k, v in d.get("secondary", {}).items():
    print(k, round(v["value"]), "frames/s", round(v["ms_per_step"], 3), "ms/step", "kernel frac", round(v["roofline"]["frac"], 4), "step frac", round(v["roofline"]["step_level"]["frac"], 4))
PY
