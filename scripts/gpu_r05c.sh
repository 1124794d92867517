#!/bin/bash
# round 5, call c: lone-wave scalar-cache prefetch (rdf_lone) A/B - small launches and the c2 / c3 throughput lines must not move
TAG=r05c; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py tests/test_zz_late_gpu.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for L in 0 -1 1; do
  echo "== rdf_lone=$L"
  for G in 1 4 16; do timeout 300 python scripts/exp_round_cost.py c2 $G rdf_lone=$L 2>&1 | tail -1 | cut -c1-600; done
  timeout 300 python scripts/exp_round_cost.py c3 1 rdf_lone=$L 2>&1 | tail -1 | cut -c1-600
done > $OUT/small_launch_ab.txt 2>&1
cat $OUT/small_launch_ab.txt
for L in 0 -1; do
  timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --opt rdf_lone=$L > $OUT/bench_c3_lone$L.json 2>> $OUT/bench.err
  timeout 300 python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --opt rdf_lone=$L > $OUT/bench_c2_lone$L.json 2>> $OUT/bench.err
  python - <<PY
import json
for w in ("c3","c2"):
    d=json.load(open("$OUT/bench_%s_lone$L.json" % w)); print(w, "rdf_lone=$L", round(d["value"]), "frames/s", {k: round(v/d["steps"],3) for k,v in d["kernel_ms"].items() if k in ("rdf_pencil","cells_build")})
PY
done
timeout 600 python scripts/exp_threads.py c2 > $OUT/call_pattern.txt 2>&1; cat $OUT/call_pattern.txt | cut -c1-600
