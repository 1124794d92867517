#!/bin/bash
# r03ar: small pair launches deal a chunk's neighbour pencils to separate work items (rdf_nsplit): A/B on the call pattern, then - the
# kernel file changed - the GPU suite, the default line, kernel stats and the PMC passes on the new source
T=${1:-r03ar}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads || exit 1
{ for o in "rdf_nsplit=0" "rdf_nsplit=-1"; do echo "## $o"; VMD_OPTS="$o" /tmp/exp_threads 100002 1000; VMD_OPTS="$o" /tmp/exp_threads 1000002 200; VMD_OPTS="$o" /tmp/exp_threads 30000 2000; done; } 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
for o in "rdf_nsplit=0" "rdf_nsplit=-1"; do for a in "c2 1" "c2 4" "c2 16"; do timeout 300 python scripts/exp_round_cost.py $a $o 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$o', d['workload'], d['frames_per_call'], 'frames per call: %.0f us per call, pair kernel %.0f us, cell build %.0f us' % (d['per_call_us'], d['rdf_pencil']['us_per'], d['cells_build']['us_per']))" ; done; done | tee $O/round_cost.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $O/prof_c3.log 2>&1
find $O/prof_c3 -name "*kernel_trace.csv" -size +5M -delete
cd $R
for w in c3 c2 c4; do
  extra="--workload $w"; [ $w = c3 ] && extra="--no-secondary"
  bash $R/scripts/gpu_pmc.sh ${T}_pmc_$w $extra > $O/pmc_$w.log 2>&1
  python $R/scripts/pmc_traffic.py $R/gpurun_out/${T}_pmc_$w $w 500 $O/pmc_traffic.json > /dev/null
done
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", "counters match (old json)", d["roofline"].get("traffic_counters_match_kernel_source"))
for k, v in d.get("secondary", {}).items(): print(k, round(v["value"]))
t = json.load(open("$O/pmc_traffic.json"))
for w in t: print(w, t[w].get("kernels_sha256_16"), {k: round(v["hbm_bytes_per_launch_read_x2"] / t[w]["frames_per_launch"] / 1e6, 3) for k, v in t[w]["kernels"].items() if k.startswith("k_rdf") or k.startswith("k_sdf_sc")})
PY
