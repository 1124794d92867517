#!/bin/bash
# round 6, pair-kernel code-size / pop A/B: parity first (the pops, the asm-vs-twin check, full sizes, fuzz seeds: the out-of-line exact path is
# behind all of them), then c3 / c2 / c5 / c3d with rdf_pop = 0 / 1 on the product build, and c3 on the experiment builds under build/ if present
T=${1:-r06p}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "pop_variants or hit_compaction or twin or full_size or fuzz or coevaluated or split_pencils" > $O/pytest_pop.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_pop.log | cut -c1-250
run() {  # label lib args
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then [ -f "$R/$lib" ] || return 0; export VIAMD_AMD_LIB=$R/$lib; else unset VIAMD_AMD_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-secondary "$@" > $O/$label.json 2> $O/$label.err || echo "$label rc=$?"
  python - $O/$label.json $label <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms/step", {k: round(v / d["steps"], 3) for k, v in d.get("kernel_ms", {}).items() if k in ("rdf_pencil", "cells_build")})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
{
for rep in 1 2; do
  for pop in 0 1; do run c3_pop${pop}_$rep "" --workload c3 --steps 10 --warmup 2 --opt rdf_pop=$pop; done
  run c3_inline_slow_path_$rep build/libviamd_inl2.so --workload c3 --steps 10 --warmup 2
  run c3_no_slow_path_WRONG_COUNTS_$rep build/libviamd_noslow.so --workload c3 --steps 10 --warmup 2
done
for pop in 0 1; do run c2_pop$pop "" --workload c2 --steps 20 --warmup 3 --opt rdf_pop=$pop; done
for pop in 0 1; do run c5_pop$pop "" --workload c5 --steps 4 --warmup 1 --opt rdf_pop=$pop; done
for pop in 0 1; do run c3d_pop$pop "" --workload c3d --steps 6 --warmup 1 --opt rdf_pop=$pop; done
} | tee $O/ab.txt
