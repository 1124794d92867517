#!/bin/bash
# r03aj: process-wide cache of device / pinned blocks, streams and events: the GPU suite with it, then what a fresh eval costs
# (VIAMD creates one per script edit) with pool_mb = 0 (every life cycle through the runtime) and with the cache
T=${1:-r03aj}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for a in "c2 device" "c2 xtc" "c4 device" "c3 device" "c5 device"; do
  for o in "pool_mb=0" ""; do timeout 600 python scripts/exp_fresh_eval.py $a $o 2>> $O/err.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); f = d['fresh_evals'][1:]
m = lambda k: sum(x[k] for x in f) / len(f)
print('%-3s %-7s %-10s reused eval %8.2f ms | fresh eval: create %6.2f + first range %8.2f + free %6.2f = %8.2f ms' % (d['workload'], d['trajectory'], '$o' or 'cache', d['reused_eval_ms'], m('create_ms'), m('first_range_ms'), m('free_ms'), m('create_ms') + m('first_range_ms') + m('free_ms')))" | tee -a $O/fresh.txt; done; done
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2>> $O/err.log
python -c "
import json; d = json.load(open('$O/bench_default.json'))
print('c3', round(d['value']), {k: round(v['value']) for k, v in d['secondary'].items()})"
grep -v amdgpu.ids $O/err.log | tail -5
