#!/bin/bash
# parity tests + all four workloads (no CPU baseline unless $2 = cpu).  usage: bash scripts/gpu_bench_all.sh <tag> [cpu]
TAG=${1:-b}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
EXTRA="--no-cpu-baseline"; [ "$2" = "cpu" ] && EXTRA=""
for w in c2 c3 c4 c5; do
  timeout 900 python bench.py --workload $w $EXTRA > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$w.json"))
    cb=d.get("cpu_baseline",{})
    print("$w", round(d["value"]), "frames/s | %.3g pairs/s |"%d["pairs_per_s"], {k:round(v,1) for k,v in d["kernel_ms"].items()}, "| roofline frac %.4f"%d["roofline"]["frac"], "| cpu", round(cb.get("value",0),2), cb.get("sample","")[:40], "| x%.0f"%d.get("gpu_over_cpu",0))
except Exception as e:
    print("$w FAILED", e); print(open("$OUT/bench_$w.err").read()[-1500:])
PY
done
