#!/bin/bash
# Round 2, GPU call 5: pair-entry compaction (variant 2) after the two-plane stack / no-self-pair changes: parity + A/B + counters
TAG=${1:-r02e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "compaction_variants or edge_cases or coevaluated" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest_gpu.log
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
ab c3_v0   --workload c3 --steps 6 --variant 0
ab c3_v2   --workload c3 --steps 6 --variant 2
ab c2_v0   --workload c2 --steps 20 --variant 0
ab c2_v2   --workload c2 --steps 20 --variant 2
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_c3v2 --no-secondary --variant 2 > $OUT/pmc_c3v2.log 2>&1
grep -A 27 "k_rdf_pencil" $R/gpurun_out/${TAG}_pmc_c3v2/summary.txt | head -30
echo done
