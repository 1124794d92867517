#!/bin/bash
# round 5, call a: sub-wave tile calibration + the c3 line of the unchanged tree for reference
TAG=r05a; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/subwave_calib scripts/subwave_calib.hip > $OUT/build.log 2>&1 || { cat $OUT/build.log; exit 1; }
timeout 600 /tmp/subwave_calib > $OUT/subwave_calib.txt 2>&1; echo "calib rc=$?"
cat $OUT/subwave_calib.txt
timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench_c3.json")); print("c3", round(d["value"]), "frames/s", d["kernel_ms"])
PY
