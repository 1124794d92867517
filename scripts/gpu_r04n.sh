#!/bin/bash
# round 4, a longer randomised campaign on the last tree: scenario fuzz (RDF / SDF / distance against the oracle), XTC decoder fuzz, filtered-evaluation fuzz,
# read-ahead stress with other seeds and sizes
T=${1:-r04n}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
for r in "8000 8400" "8400 8800" "8800 9200"; do echo "# python scripts/fuzz_gpu.py $r 20"; timeout 1200 python scripts/fuzz_gpu.py $r 20 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "# python scripts/fuzz_xtc.py 600 41 gpu"; timeout 900 python scripts/fuzz_xtc.py 600 41 gpu 2>&1 | grep -v amdgpu.ids | tail -2
echo "# python scripts/fuzz_emu_filtered.py 120 13 gpu"; timeout 900 python scripts/fuzz_emu_filtered.py 120 13 gpu 2>&1 | grep -v amdgpu.ids | tail -2
B="-std=c++17 -O2 -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread"
g++ tests/native/stress_readahead.cpp $B -o /tmp/stress_ra && for a in "800 240 30000 51" "800 240 30000 52" "400 600 100002 53" "60 96 1000002 54" "1500 64 3000 55"; do echo "# stress_readahead $a"; timeout 1200 /tmp/stress_ra $a 2>&1 | grep -v amdgpu.ids | tail -1; done
} | tee $O/campaign.txt
