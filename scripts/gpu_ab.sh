#!/bin/bash
# A/B of alternative builds of the library (VIAMD_AMD_LIB) x launch options on the default bench workload.
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_ab.sh <tag> "<lib-or-empty>|<bench args>" ...'
TAG=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
i=0
for spec in "$@"; do
  i=$((i+1))
  lib=${spec%%|*}; args=${spec#*|}
  echo "== [$i] lib=${lib:-default} args=$args"
  if [ -n "$lib" ]; then export VIAMD_AMD_LIB=$R/$lib; else unset VIAMD_AMD_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline $args > $OUT/ab_$i.json 2> $OUT/ab_$i.err
  python - $OUT/ab_$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["unit"], {k: round(v, 1) for k, v in d.get("kernel_ms", {}).items()})
except Exception as e:
    print("FAILED", e)
PY
done
