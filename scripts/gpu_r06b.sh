#!/bin/bash
# round 6, call B: the -m gpu suite on the split tree, smoke, the default line (c4 without the zeroing pass, c1, c3_125 / c5_125, CPU median),
# PMC of c5 / c3 with the timed-region marker (steady-state cell-build traffic), the call pattern from C++ pool threads
T=${1:-r06b}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step frac", round(d["roofline"]["frac"], 4), "cpu", round(d["cpu_baseline"]["value"], 1), [round(x, 1) for x in d["cpu_baseline"]["samples"]], "gpu/cpu", round(d["gpu_over_cpu"], 1))
print("launches/step", d["kernel_launches_per_step"], "merge payload", d.get("merge_payload_bytes"))
for k, v in d.get("secondary", {}).items():
    if k == "c1": print("c1", json.dumps({a: v[a] for a in ("gpu_ms", "cpu_ms", "gpu_over_cpu", "work_pairs_times_frames")})[:600]); continue
    print(k, round(v["value"]), round(v["ms_per_step"], 3), v.get("strong_scaling_bound_8_gpus"), v.get("rank_part_ms"), v.get("merge_payload_bytes"), v.get("kernel_launches_per_step"))
PY
for w in c5 c3; do
  extra="--workload $w --no-secondary"
  bash $R/scripts/gpu_pmc.sh ${T}_pmc_$w $extra > $O/pmc_$w.log 2>&1
  python $R/scripts/pmc_traffic.py $R/gpurun_out/${T}_pmc_$w $w 1000 $O/pmc_traffic.json 2 1000 > /dev/null
  cp $R/gpurun_out/${T}_pmc_$w/summary.txt $O/pmc_summary_$w.txt 2>/dev/null
  rm -rf $R/gpurun_out/${T}_pmc_$w
done
python - <<PY
import json
d = json.load(open("$O/pmc_traffic.json"))
for w in ("c5", "c3"):
    print(w, "timed region only:", d[w].get("timed_region_only"))
    for k, v in d[w]["kernels"].items():
        if k.startswith("k_cells") or k.startswith("k_rdf"):
            print("  ", k, "disp/step", v["dispatches_per_step"], "GB/step (2xFETCH+WRITE)", round(v.get("hbm_bytes_per_step_read_x2", 0) / 1e9, 2), "write GB/step", round(v["write_bytes_per_launch"] * v["dispatches_per_step"] / 1e9, 2))
PY
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads && { /tmp/exp_threads 100002 1000; VMD_SDF=1 /tmp/exp_threads 100002 10000; } > $O/readahead_call_pattern.txt 2>&1; cat $O/readahead_call_pattern.txt | cut -c1-300 | tail -12
du -sh $O
