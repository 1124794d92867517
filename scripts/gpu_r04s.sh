#!/bin/bash
# round 4, last evidence call (the PMC passes run BEFORE the default line, so that the line reads counters of the kernel source it runs)
# round 4 evidence on the final kernel source: the whole -m gpu suite, smoke, driver-style default line, rocprofv3 kernel stats of the same
# command, PMC passes at the bench's own frames per launch (c3 / c2 / c5: 1 000, c4: 10 000), the call pattern, the 2-rank shared-GPU line
T=${1:-r04s}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > $O/device.txt
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for w in c3 c2 c4 c5; do
  extra="--workload $w"; [ $w = c3 ] && extra="--no-secondary"
  bash $R/scripts/gpu_pmc.sh ${T}_pmc_$w $extra > $O/pmc_$w.log 2>&1
  fpl=1000; [ $w = c4 ] && fpl=10000
  python $R/scripts/pmc_traffic.py $R/gpurun_out/${T}_pmc_$w $w $fpl $O/pmc_traffic.json > /dev/null
  cp $R/gpurun_out/${T}_pmc_$w/summary.txt $O/pmc_summary_$w.txt 2>/dev/null
  rm -rf $R/gpurun_out/${T}_pmc_$w            # the per-dispatch counter tables: tens of MB per workload, summarised above (gpurun returns 64 MiB at most)
done
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json      # the default line below prices its traffic against counters of THIS kernel source
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -o c3 -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $O/prof_c3.log 2>&1
find $O/prof_c3 -name "*kernel_trace.csv" -delete
find $O/prof_c3 -name "*agent_info.csv" -delete
cd $R
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step frac", round(d["roofline"]["frac"], 4), "cpu", round(d["cpu_baseline"]["value"], 1), "gpu/cpu", round(d["gpu_over_cpu"], 1))
for k, v in d.get("secondary", {}).items(): print(k, round(v["value"]), round(v["ms_per_step"], 3), "kernel frac", round(v["roofline"]["frac"], 4), "step frac", round(v["roofline"]["step_level"]["frac"], 4))
t = json.load(open("$O/pmc_traffic.json"))
for w in t: print(w, t[w].get("kernels_sha256_16"), t[w]["frames_per_launch"], {k: round(v["hbm_bytes_per_launch_read_x2"] / t[w]["frames_per_launch"] / 1e6, 3) for k, v in t[w]["kernels"].items() if k.startswith("k_rdf") or k.startswith("k_sdf_sc") or k.startswith("k_cells")})
PY
g++ -std=c++17 -O2 tests/native/exp_threads.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/exp_threads && { /tmp/exp_threads 100002 1000; VMD_SDF=1 /tmp/exp_threads 100002 10000; } 2>&1 | grep -v amdgpu.ids | tee $O/threads.txt
VIAMD_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 2> $O/bench_share2.err | grep "^{" > $O/bench_share2.json; echo "share2 rc=$?"
du -sh $R/gpurun_out
