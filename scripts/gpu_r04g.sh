#!/bin/bash
# round 4 robustness: the seeded random scenarios against the oracle on the product build (new seeds), filtered-evaluation fuzz, the read-ahead
# stress at three system sizes, the evaluator life-cycle stress - everything that changed this round under load
T=${1:-r04g}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "# python scripts/fuzz_gpu.py 7000 7300 20"
timeout 900 python scripts/fuzz_gpu.py 7000 7300 20 2>&1 | grep -v amdgpu.ids | tail -5
echo "# python scripts/fuzz_emu_filtered.py 60 9 gpu"
timeout 600 python scripts/fuzz_emu_filtered.py 60 9 gpu 2>&1 | grep -v amdgpu.ids | tail -3
B="-std=c++17 -O2 -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread"
g++ tests/native/stress_readahead.cpp $B -o /tmp/stress_ra && {
  echo "# stress_readahead 600 240 30000 11   (iterations frames atoms seed)"; timeout 900 /tmp/stress_ra 600 240 30000 11 2>&1 | grep -v amdgpu.ids | tail -2
  echo "# stress_readahead 300 1000 100002 12"; timeout 900 /tmp/stress_ra 300 1000 100002 12 2>&1 | grep -v amdgpu.ids | tail -2
  echo "# stress_readahead 40 96 1000002 13"; timeout 900 /tmp/stress_ra 40 96 1000002 13 2>&1 | grep -v amdgpu.ids | tail -2
}
g++ tests/native/stress_eval.cpp $B -o /tmp/stress_eval && { echo "# stress_eval 600 48"; AMD_LOG_LEVEL=1 timeout 900 /tmp/stress_eval 600 48 2>&1 | grep -v amdgpu.ids | tail -2; }
} | tee $O/robustness.txt
