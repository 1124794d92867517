#!/bin/bash
# round 5, call b: the GPU suite on the tree so far (shim decorator, read-ahead fixes), the one-frame launch under counters,
# c4's step split at 10 000 and 1 250 frames
TAG=r05b; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/one/p$i -o p$i -- python $R/scripts/exp_one_frame_pmc.py > $OUT/one_p$i.log 2>&1
  echo "one-frame pass $i ($line) rc=$?"
done <<'PASSES'
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES
PASSES
cd $R
python scripts/exp_one_frame_pmc.py summarize $OUT/one > $OUT/one_frame_pmc.txt 2>&1; cat $OUT/one_frame_pmc.txt
find $OUT/one -name "*agent_info.csv" -delete
for F in 10000 1250; do
  timeout 300 python bench.py --workload c4 --frames $F --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4_$F.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_c4_$F.json")); print("c4 $F frames:", round(d["ms_per_step"],3), "ms/step", {k: round(v/d["steps"],4) for k,v in d["kernel_ms"].items()})
PY
done
tail -3 $OUT/bench.err
