#!/bin/bash
# r03y: persistent streaming SDF scatter (sdf_wave = 2 / n blocks) against the one-shot kernels on c4; parity first
T=${1:-r03y}; O=gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sdf" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() { tag=$1; shift
  timeout 300 python bench.py --workload c4 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 "$@" 2>> $O/err.log | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernel_ms']; s = d['steps']
print('%-28s %9.0f frames/s  %.3f ms/step  scatter %.3f ms  align %.3f ms  kernel frac %.4f  step frac %.4f' % ('$tag', d['value'], d['ms_per_step'], k['sdf_scatter'] / s, k['sdf_align'] / s, d['roofline']['frac'], d['roofline']['step_level']['frac']))" | tee -a $O/ab.txt
}
run default
run wave1 --opt sdf_wave=1
run stream_2048 --opt sdf_wave=2
run stream_1024 --opt sdf_wave=1024
run stream_512 --opt sdf_wave=512
run stream_4096 --opt sdf_wave=4096
run stream_2048_ilp8 --opt sdf_wave=2 --opt sdf_ilp=8
run stream_1024_ilp8 --opt sdf_wave=1024 --opt sdf_ilp=8
run default_again
grep -v amdgpu.ids $O/err.log | tail -5
