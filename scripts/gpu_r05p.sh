T=r05z; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$T; mkdir -p $O; cd $R; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cp profiles/pmc_traffic.json $O/pmc_traffic.json
bash $R/scripts/gpu_pmc.sh ${T}_pmc_c3d --workload c3d > $O/pmc_c3d.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${T}_pmc_c3d c3d 200 $O/pmc_traffic.json 2 200 > /dev/null
cp $R/gpurun_out/${T}_pmc_c3d/summary.txt $O/pmc_summary_c3d.txt; rm -rf $R/gpurun_out/${T}_pmc_c3d
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json")); s = d["secondary"]
print("c3", round(d["value"]), round(d["gpu_over_cpu"], 1), d["fractions_within_0_1"], d["roofline"]["traffic_counters_match_kernel_source"])
print("c3d", round(s["c3d"]["value"]), s["c3d"]["roofline"]["frac"], s["c3d"]["roofline"]["valu"] and {k: round(v, 3) for k, v in s["c3d"]["roofline"]["valu"].items() if k in ("frac", "busy")}, s["c3d"]["roofline"]["traffic"])
for k in ("c2", "c4", "c5", "c4_1250"): print(k, round(s[k]["value"]), round(s[k]["ms_per_step"], 3))
PY
