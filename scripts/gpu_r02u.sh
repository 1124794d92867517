#!/bin/bash
# Round 2: cell build A/B (12-byte records: r02u; block-local sort in level 1: r02x)
TAG=${1:-r02x}
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
    print("%-20s %10.0f frames/s %9.3f ms/step  hits/step %d  kernels %s" % ("$name", d["value"], d["ms_per_step"], round(d["pairs_per_s"] * d["ms_per_step"] / 1e3), {k: round(v, 2) for k, v in d["kernel_ms"].items()}))
except Exception as ex:
    print("$name FAILED", ex)
PY
}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py -x -q -k "cell_build or random_rdf or overflow" 2>&1 | tail -2
for v in 0 1 0 1; do ab c3_binlds_$v --workload c3 --steps 6 --opt cells_bin_lds=$v; done
for v in 0 1; do ab c2_binlds_$v --workload c2 --steps 20 --opt cells_bin_lds=$v; done
for v in 0 1; do ab c5_binlds_$v --workload c5 --steps 3 --opt cells_bin_lds=$v; done
tail -3 $OUT/ab.err
echo done
