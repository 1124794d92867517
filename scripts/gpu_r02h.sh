#!/bin/bash
# Round 2: shared LDS histogram (8 waves/SIMD) A/B, aux-stream clearing DMA for c4
TAG=${1:-r02h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "compaction_variants or sdf or filtered" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
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
ab c3_base    --workload c3 --steps 6
ab c3_shist   --workload c3 --steps 6 --opt rdf_shared_hist=1
ab c2_base    --workload c2 --steps 20
ab c2_shist   --workload c2 --steps 20 --opt rdf_shared_hist=1
ab c5_base    --workload c5 --steps 3
ab c5_shist   --workload c5 --steps 3 --opt rdf_shared_hist=1
ab c3_shist_blocks  --workload c3 --steps 6 --opt rdf_shared_hist=1 --opt rdf_blocks=2048
ab c4         --workload c4 --steps 5
ab c4_20      --workload c4 --steps 20
echo done
