#!/bin/bash
# Round 2: fine-cell size sweep (the x window of a segment is rounded outward to whole fine cells)
TAG=${1:-r02l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
ab() {
  local name=$1; shift
  timeout 600 python bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/ab_$name.json 2>> $OUT/ab.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_$name.json"))
    print("%-28s %12.0f frames/s %9.3f ms/step  kernels %s" % ("$name", d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["kernel_ms"].items()}))
except Exception as ex:
    print("$name FAILED", ex)
PY
}
for dv in 8 12 16 24 32; do ab c3_nxf$dv --workload c3 --steps 5 --opt nxf_divisor=$dv; done
for dv in 8 16 32; do ab c2_nxf$dv --workload c2 --steps 20 --opt nxf_divisor=$dv; done
for dv in 8 16; do ab c5_nxf$dv --workload c5 --steps 3 --opt nxf_divisor=$dv; done
tail -3 $OUT/ab.err
echo done
