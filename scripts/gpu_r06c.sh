#!/bin/bash
# round 6, call C: why does c5 build cells 3.5 times per batch (exp_c5_builds.py)?  A/B of the computed selection index (cells_sel_pattern) on c3 / c5
T=${1:-r06c}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "cell or rdf" > $O/pytest_cells.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_cells.log
timeout 600 python scripts/exp_c5_builds.py 1000 > $O/c5_builds.txt 2>&1; cat $O/c5_builds.txt | tail -8
for v in 1 0; do
  timeout 600 python bench.py --workload c3 --no-secondary --no-cpu-baseline --steps 10 --warmup 3 --opt cells_sel_pattern=$v > $O/c3_pattern_$v.json 2> $O/c3_pattern_$v.err
  timeout 600 python bench.py --workload c5 --no-secondary --no-cpu-baseline --steps 3 --warmup 2 --opt cells_sel_pattern=$v > $O/c5_pattern_$v.json 2> $O/c5_pattern_$v.err
done
python - <<PY
import json
for w in ("c3", "c5"):
    for v in (1, 0):
        d = json.load(open(f"$O/{w}_pattern_{v}.json"))
        print(w, "cells_sel_pattern", v, round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms/step; cells_build ms/step", round(d["kernel_ms"]["cells_build"] / d["steps"], 3), d.get("cell_build", {}).get("launches_per_step"), d.get("cell_build", {}).get("bucket_overflows_since_creation"))
PY
