#!/bin/bash
# round 6, the ONE evidence call on the final tree: the whole -m gpu suite, smoke, PMC passes of every workload at the bench's
# own frames per step - timed region only (k_marker_timed_region) and BEFORE the default line, so that the line replays counters of the kernel
# source it runs -, the driver-style default line, rocprofv3 kernel stats of every workload's command, VIAMD's call pattern from C++ pool
# threads, the 2-rank line on one shared GPU
T=${1:-r06z}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd $R
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > $O/device.txt
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for w in c3 c2 c4 c5 c3d; do
  extra="--workload $w --no-secondary"
  bash $R/scripts/gpu_pmc.sh ${T}_pmc_$w $extra > $O/pmc_$w.log 2>&1
  fps=1000; [ $w = c4 ] && fps=10000; [ $w = c3d ] && fps=200
  # timed steps in the profiled run: --steps 2 (the warm-up step lies in front of the marker and is dropped); frames per step = the workload's own
  python $R/scripts/pmc_traffic.py $R/gpurun_out/${T}_pmc_$w $w $fps $O/pmc_traffic.json 2 $fps > /dev/null
  cp $R/gpurun_out/${T}_pmc_$w/summary.txt $O/pmc_summary_$w.txt 2>/dev/null
  rm -rf $R/gpurun_out/${T}_pmc_$w
done
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd /tmp
for w in c3 c2 c4 c5 c3d; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o $w -- python $R/bench.py --workload $w --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $O/prof_$w.log 2>&1
  find $O/prof_$w -name "*kernel_trace.csv" -delete; find $O/prof_$w -name "*agent_info.csv" -delete
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_$w.csv
  rm -rf $O/prof_$w
done
cd $R
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step frac", round(d["roofline"]["frac"], 4), "valu", {k: round(v, 3) for k, v in d["roofline"]["valu"].items() if k in ("frac", "busy")}, "cpu", round(d["cpu_baseline"]["value"], 1), [round(x, 1) for x in d["cpu_baseline"]["samples"]], "gpu/cpu", round(d["gpu_over_cpu"], 1), "fractions ok", d["fractions_within_0_1"], "counters current", d["roofline"]["traffic_counters_match_kernel_source"])
print("cell build", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["cell_build"].items() if k != "traffic_source"})
for k, v in d.get("secondary", {}).items():
    if k == "c1": print("c1", json.dumps({a: v[a] for a in ("gpu_ms", "cpu_ms", "gpu_over_cpu", "work_pairs_times_frames")})[:700]); continue
    print(k, round(v["value"]), round(v["ms_per_step"], 3), {a: (round(b["frac"], 4), round(b["step_level"]["frac"], 4)) for a, b in v.items() if a == "roofline"}, v.get("strong_scaling_bound_8_gpus"), v.get("rank_part_ms"), (v.get("cell_build") or {}).get("launches_per_step"), (v.get("cell_build") or {}).get("traffic_ratio"))
PY
for w in c3 c4; do echo "== kernel stats $w"; head -5 $O/rocprofv3_kernel_stats_$w.csv | cut -c1-170; done
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads && { /tmp/exp_threads 100002 1000; VMD_SDF=1 /tmp/exp_threads 100002 10000; } > $O/readahead_call_pattern.txt 2>&1; cat $O/readahead_call_pattern.txt | cut -c1-400
VIAMD_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 2> $O/bench_share2.err | grep "^{" > $O/bench_share2.json; echo "share2 rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_share2.json")); print("share2: n_gpus", d["n_gpus"], "ranks", d.get("ranks"), round(d["value"]), "frames/s", d["config"]["parallelism"][:60])
PY
du -sh $R/gpurun_out/$T
