#!/bin/bash
# quick iteration: parity tests + bench c2/c3 (+ optional PMC pass).  usage: bash scripts/gpu_quick.sh <tag> [pmc]
TAG=${1:-q}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --steps 10 > $OUT/bench_c2.json 2> $OUT/bench.err; python - <<PY
import json
for n in ("bench_c2",):
    d=json.load(open("$OUT/"+n+".json")); print(n, round(d["value"]), "frames/s", "%.3g pairs/s"%d["pairs_per_s"], d["kernel_ms"], d["roofline"]["avg_launch_ms"], d["roofline"]["frames_per_launch"])
PY
timeout 300 python bench.py --workload c3 --frames 250 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c3.json 2>> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench_c3.json")); print("c3", round(d["value"]), "frames/s", "%.3g pairs/s"%d["pairs_per_s"], d["kernel_ms"])
PY
tail -3 $OUT/bench.err
if [ "$2" = "pmc" ]; then
  cd /tmp
  for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --frames 500 > $OUT/p$i.log 2>&1
  done
  python $R/scripts/pmc_summary.py $OUT | grep -A20 "k_rdf_pencil"
  find $OUT -name "*kernel_trace.csv" -size +5M -delete
fi
