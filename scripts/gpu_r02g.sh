#!/bin/bash
# Round 2: re-validation after the pinned-view change: full GPU suite three times, smoke, default bench.
TAG=${1:-r02g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for i in 1 2; do
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu_$i.log 2>&1; rc=$?
  echo "pytest run $i rc=$rc: $(grep -E 'passed|failed' $OUT/pytest_gpu_$i.log | tail -1)"
  [ $rc -ne 0 ] && { grep -v "^$" $OUT/pytest_gpu_$i.log | grep -v 'File "/usr' | tail -20; }
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step", "valu", d["roofline"]["valu"] and round(d["roofline"]["valu"]["frac"], 3))
for k, v in d.get("secondary", {}).items():
    print(k, round(v["value"]), "frames/s", round(v["ms_per_step"], 3), "ms/step", "kernel frac", round(v["roofline"]["frac"], 4), "step frac", round(v["roofline"]["step_level"]["frac"], 4))
PY
echo done
