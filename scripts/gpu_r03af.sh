#!/bin/bash
# r03af: end-of-round robustness on the restructured batch loop: 2 x 1000 evaluator life cycles (SDMA on / off), more fuzz seeds,
# the GPU suite a second and third time (flakiness)
T=${1:-r03af}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
g++ -std=c++17 -O2 tests/native/stress_eval.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/stress_eval || exit 1
{ for sdma in 1 0; do
  t0=$(date +%s.%N)
  HSA_ENABLE_SDMA=$sdma AMD_LOG_LEVEL=1 timeout 900 /tmp/stress_eval 1000 48 > $O/stress_sdma$sdma.out 2> $O/stress_sdma$sdma.err; rc=$?
  t1=$(date +%s.%N)
  echo "HSA_ENABLE_SDMA=$sdma rc=$rc $(tail -1 $O/stress_sdma$sdma.out) wall=$(echo "$t1 - $t0" | bc) s, stderr lines: $(wc -l < $O/stress_sdma$sdma.err)"
done
echo "# python scripts/fuzz_gpu.py 6000 6120 20"; timeout 900 python scripts/fuzz_gpu.py 6000 6120 20 2>&1 | grep -v amdgpu.ids | tail -3
echo "# python scripts/fuzz_gpu.py 7000 7020 60"; timeout 900 python scripts/fuzz_gpu.py 7000 7020 60 2>&1 | grep -v amdgpu.ids | tail -3
echo "# python scripts/fuzz_emu_filtered.py 40 9 gpu"; timeout 900 python scripts/fuzz_emu_filtered.py 40 9 gpu 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|^ *|" | tail -2 | cut -c1-300
echo "# python scripts/fuzz_xtc.py 300 33 gpu"; timeout 900 python scripts/fuzz_xtc.py 300 33 gpu 2>&1 | grep -v "amdgpu.ids\|warning\|^ *[0-9]* |\|^ *|" | tail -1 | cut -c1-300
for k in 2 3; do timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu_$k.log 2>&1; echo "pytest -m gpu, run $k: rc=$? $(tail -1 $O/pytest_gpu_$k.log)"; done
} | tee $O/robustness.txt
