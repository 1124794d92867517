#!/bin/bash
# Round 2: SDF scatter with per-wave compaction (no block barrier), A/B on c4 and c5
TAG=${1:-r02v}
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
    print("%-20s %10.0f frames/s %9.3f ms/step  voxel hits/step %d  kernels %s" % ("$name", d["value"], d["ms_per_step"], round(d["voxel_hits_per_s"] * d["ms_per_step"] / 1e3), {k: round(v, 2) for k, v in d["kernel_ms"].items()}))
except Exception as ex:
    print("$name FAILED", ex)
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py tests/test_full_size_gpu.py -x -q -k "sdf or config4" 2>&1 | tail -2
for v in "0 4" "1 4" "1 8" "0 4" "1 4"; do set -- $v; ab c4_wave$1_ilp$2 --workload c4 --steps 10 --opt sdf_wave=$1 --opt sdf_ilp=$2; done
tail -3 $OUT/ab.err
echo done
