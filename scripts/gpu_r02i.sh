#!/bin/bash
# Round 2: row-streaming SDF scatter A/B (c4, c5) + the new tests
TAG=${1:-r02i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "sdf or overflow or running_source or coevaluated or brute" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest_gpu.log
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
ab c4_base   --workload c4 --steps 10
ab c4_rows1  --workload c4 --steps 10 --opt sdf_rows=1
ab c4_rows2  --workload c4 --steps 10 --opt sdf_rows=2
ab c4_rows4  --workload c4 --steps 10 --opt sdf_rows=4
ab c4_base2  --workload c4 --steps 10
ab c3_1batch --workload c3 --steps 6
ab c3_2batch --workload c3 --steps 6 --batch 500
echo done
