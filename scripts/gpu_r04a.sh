#!/bin/bash
# round 4, first call: the multi-process path on one GPU (bare `bench.py --gpus 2`, host-staged collective), the new -m gpu tests,
# the call-pattern baseline (exp_threads) before read-ahead
T=${1:-r04a}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_dist.py tests/test_native.py -m gpu -x -q -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "pytest(new) rc=$?"; tail -3 $O/pytest_new.log
echo "== bare command, 2 ranks sharing GPU 0 (gloo, staged), c3 200 frames per rank"
VIAMD_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --frames 200 > $O/bench_share2.json 2> $O/bench_share2.err; echo "rc=$?"
python - <<PY
import json
try:
    d = json.load(open("$O/bench_share2.json"))
    print("n_gpus", d["n_gpus"], round(d["value"]), "frames/s", d["per_rank_ms_per_step"], d["merge"])
except Exception as e:
    print("no line:", e); print(open("$O/bench_share2.err").read()[-2000:])
PY
echo "== bare command, 2 ranks sharing GPU 0, default (secondary strong lines c4/c5)"
VIAMD_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_share2_full.json 2> $O/bench_share2_full.err; echo "rc=$?"
python - <<PY
import json
try:
    d = json.load(open("$O/bench_share2_full.json"))
    print("n_gpus", d["n_gpus"], round(d["value"]), "frames/s", d["per_rank_ms_per_step"]); print(json.dumps(d.get("secondary"))[:1500])
except Exception as e:
    print("no line:", e); print(open("$O/bench_share2_full.err").read()[-2000:])
PY
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads || exit 1
{ /tmp/exp_threads 100002 1000; VMD_SDF=1 /tmp/exp_threads 100002 10000; } 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step", "frac", round(d["roofline"]["frac"], 4), "cpu", round(d["cpu_baseline"]["value"], 1))
for k, v in d.get("secondary", {}).items():
    print(k, round(v["value"]), "frames/s", round(v["ms_per_step"], 3), "ms/step", "kernel frac", round(v["roofline"]["frac"], 4), "step frac", round(v["roofline"]["step_level"]["frac"], 4))
PY
