#!/bin/bash
# Round 2: pair-kernel tile shape A/B (split pencils) + build/pair overlap experiment
TAG=${1:-r02m}
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
    print("%-28s %12.0f frames/s %9.3f ms/step  hits/step %d  kernels %s" % ("$name", d["value"], d["ms_per_step"], round(d["pairs_per_s"] * d["ms_per_step"] / 1e3), {k: round(v, 2) for k, v in d["kernel_ms"].items()}))
except Exception as ex:
    print("$name FAILED", ex)
PY
}
timeout 300 python -m pytest tests/test_full_size_gpu.py -x -q -k "c3" 2>&1 | tail -2
for sp in "1 1" "2 1" "1 2" "2 2"; do set -- $sp; ab c3_split$1$2 --workload c3 --steps 4 --opt pencil_split_y=$1 --opt pencil_split_z=$2; done
for sp in "1 1" "2 1" "2 2"; do set -- $sp; ab c2_split$1$2 --workload c2 --steps 10 --opt pencil_split_y=$1 --opt pencil_split_z=$2; done
for sp in "1 1" "2 1"; do set -- $sp; ab c5_split$1$2 --workload c5 --steps 2 --opt pencil_split_y=$1 --opt pencil_split_z=$2; done
tail -3 $OUT/ab.err
timeout 900 python scripts/exp_overlap.py c3 2>&1 | tee $OUT/overlap_c3.txt | tail -8
echo done
