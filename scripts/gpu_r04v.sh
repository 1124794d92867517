#!/bin/bash
# stress_readahead with interrupts inside the side-by-side scenario (filtered eval interrupted and restarted; full eval interrupted, cleared and
# restarted while the filtered one carries on), MI355X
T=${1:-r04v}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
L="$R/viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread"
g++ -std=c++17 -O2 $R/tests/native/stress_readahead.cpp -I$R/include $L -o /tmp/stress_ra || exit 1
{
for s in 31 32 33 34; do timeout 100 /tmp/stress_ra 400 240 30000 $s 2>&1 | grep -v amdgpu.ids | tail -2; echo "rc=$? (seed $s)"; done
for s in 35 36; do timeout 100 /tmp/stress_ra 150 240 30000 $s sdf 2>&1 | grep -v amdgpu.ids | tail -2; echo "rc=$? (sdf, seed $s)"; done
} > $O/stress.txt 2>&1
cat $O/stress.txt
