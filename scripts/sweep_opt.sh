#!/bin/bash
# usage: bash scripts/sweep_opt.sh <workload> <key> v1 v2 ...   (bench.py --opt key=v, prints frames/s and kernel ms)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
W=$1; K=$2; shift 2
for v in "$@"; do
  python bench.py --workload $W --no-cpu-baseline --steps 6 --opt $K=$v 2>/dev/null > /tmp/sweep.json
  python - <<PY
import json
d=json.load(open("/tmp/sweep.json")); print("$K=$v", round(d["value"]), {k:round(x,1) for k,x in d["kernel_ms"].items()})
PY
done
