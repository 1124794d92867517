#!/bin/bash
# c4 with every SDF scatter variant (bench.py --opt): usage on the GPU box: bash scripts/sweep_sdf.sh   (round 5: the defaults - ilp 4, block kernel - are still the fastest, profiles/r05w_campaign.txt)
cd $GRAFT_REPO_ROOT
for O in "sdf_ilp=4" "sdf_ilp=8" "sdf_ilp=16" "sdf_nt=1" "sdf_rows=1" "sdf_rows=2" "sdf_rows=4" "sdf_wave=1" "sdf_wave=2" "sdf_ilp=4"; do
  python bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --opt $O 2>/dev/null > /tmp/s.json
  python - <<PY
import json
d=json.load(open("/tmp/s.json")); print("$O", round(d["ms_per_step"],3), "ms/step, scatter", round(d["roofline"]["avg_launch_ms"],3), "ms, frac", round(d["roofline"]["frac"],3))
PY
done
