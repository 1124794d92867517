#!/bin/bash
T=${1:-r03as}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step", "frac", round(d["roofline"]["frac"], 4), "counters match", d["roofline"].get("traffic_counters_match_kernel_source"), "traffic", d["roofline"].get("traffic"))
for k, v in d.get("secondary", {}).items(): print(k, round(v["value"]), round(v["roofline"]["step_level"]["frac"], 4))
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "half_shell or overflow or pool_threads" 2>&1 | tail -2
