#!/bin/bash
# Round 2, GPU call 4: pair-entry hit compaction (rdf_variant 2) - parity tests, then A/B against variant 0; SDF scatter ILP;
# cell build after the ILP-8 / single-read changes.
TAG=${1:-r02d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 $OUT/pytest_gpu.log
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
echo "== A/B"
ab c3_v0   --workload c3 --steps 6 --variant 0
ab c3_v2   --workload c3 --steps 6 --variant 2
ab c2_v0   --workload c2 --steps 20 --variant 0
ab c2_v2   --workload c2 --steps 20 --variant 2
ab c5_v0   --workload c5 --steps 3 --variant 0
ab c5_v2   --workload c5 --steps 3 --variant 2
ab c3d_v0  --workload c3d --steps 2 --variant 0
ab c3d_v2  --workload c3d --steps 2 --variant 2
ab c4_ilp4  --workload c4 --steps 5
ab c4_ilp8  --workload c4 --steps 5 --opt sdf_ilp=8
ab c4_ilp16 --workload c4 --steps 5 --opt sdf_ilp=16
tail -5 $OUT/ab.err
echo "== PMC c3 variant 2 (instruction counts)"
bash $R/scripts/gpu_pmc.sh ${TAG}_pmc_c3v2 --no-secondary --variant 2 > $OUT/pmc_c3v2.log 2>&1
python $R/scripts/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_c3v2 c3 500 $OUT/pmc_traffic.json | tail -12
grep -A 27 "k_rdf_pencil" $R/gpurun_out/${TAG}_pmc_c3v2/summary.txt | head -30
echo done
