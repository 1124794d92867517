#!/bin/bash
# small pair launches touch their lines first (k_rdf_pencil<.., PF>, option rdf_prefetch: -1 automatic, 0 off, 1 always): A/B on single small calls
# and on the lone-caller pattern; the parity test of the instantiation; the default line (large launches take the plain instantiation)
T=${1:-r04h}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "prefetch or compaction or open_bound or triclinic" > $O/pytest_pf.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_pf.log
for o in "rdf_prefetch=0" "rdf_prefetch=-1" "rdf_prefetch=1"; do for a in "c2 1" "c2 4" "c2 16" "c2 64" "c3 1" "c3 4"; do timeout 300 python scripts/exp_round_cost.py $a $o 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$o', d['workload'], d['frames_per_call'], 'frames per call: %.0f us per call, pair kernel %.0f us, cell build %.0f us' % (d['per_call_us'], d['rdf_pencil']['us_per'], d['cells_build']['us_per']))" ; done; done | tee $O/round_cost.txt
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads || exit 1
{ for o in "rdf_prefetch=0" "rdf_prefetch=-1"; do echo "## $o"; VMD_OPTS="$o" /tmp/exp_threads 100002 1000; VMD_OPTS="$o" /tmp/exp_threads 30000 2000; done; } 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "counters match", d["roofline"]["traffic_counters_match_kernel_source"])
for k, v in d.get("secondary", {}).items(): print(k, round(v["value"]), round(v["ms_per_step"], 3))
PY
