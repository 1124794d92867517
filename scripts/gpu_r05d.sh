#!/bin/bash
# round 5, call d: pencil_split by density - the sub-wave model's side result (today's kernel on half-width pencils: 0.906 x at rho = 0.1)
TAG=r05d; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
for W in c3d c5; do
for S in "1 1" "2 1" "1 2" "2 2"; do
  set -- $S
  timeout 400 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --opt pencil_split_y=$1 --opt pencil_split_z=$2 > $OUT/bench_${W}_$1$2.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_${W}_$1$2.json")); print("$W split $1 x $2:", round(d["value"],1), "frames/s", {k: round(v/d["steps"],2) for k,v in d["kernel_ms"].items() if k in ("rdf_pencil","cells_build")}, "lanes/pair", round(d["pair_kernel_columns"]["candidate_lanes_per_ordered_pair"],2), "hits", d["pairs_per_s"]/d["value"])
PY
done; done 2>&1 | tee $OUT/pencil_split_by_density.txt
tail -2 $OUT/bench.err
