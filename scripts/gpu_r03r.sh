#!/bin/bash
# r03r: TRR / DCD files: frames DMA'd out of the mapped file + k_raw_f32 against load_frame on host threads
T=${1:-r03r}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_zzz_xdr_gpu.py tests/test_xdr.py tests/test_io.py -m gpu -x -q > $O/pytest_xdr.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_xdr.log
for fmt in trr dcd; do for dev in 0 1; do
  tag=${fmt}_dev$dev
  timeout 600 python bench.py --workload c2 --traj $fmt --no-cpu-baseline --steps 5 --warmup 2 --opt raw_f32_device=$dev > $O/bench_$tag.json 2>> $O/err.log
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
k=d['kernel_ms']; s=d['steps']
print('$tag', round(d['value']), 'frames/s', {a: round(b/s,2) for a,b in k.items() if not a.startswith('host_q')})
PY
done; done
timeout 600 python bench.py --workload c2 --traj pinned --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_pinned.json 2>> $O/err.log
python -c "import json;d=json.loads([l for l in open('$O/bench_pinned.json') if l.startswith('{')][-1]);print('pinned floats', round(d['value']))"
tail -3 $O/err.log
