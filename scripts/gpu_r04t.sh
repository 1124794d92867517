#!/bin/bash
# round 4, side-by-side campaign on the MI355X: what ThreadSanitizer on the emulator cannot see is the ordering of device work across the
# streams of evals that run at the same time (the emulator executes every launch synchronously).  Independent evaluations at once, the filtered
# evaluation beside its source (C ABI and md_script_* shim), ranks sharing the device - long runs, every result compared bit for bit.
T=${1:-r04t}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 AMD_LOG_LEVEL=1
L="$R/viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread"
g++ -std=c++17 -O2 $R/tests/native/concurrent_evals.cpp -I$R/include $L -o /tmp/concurrent_evals || exit 1
g++ -std=c++17 -O2 $R/tests/native/stress_readahead.cpp -I$R/include $L -o /tmp/stress_ra || exit 1
g++ -std=c++17 -O2 $R/tests/native/shim_callsites.cpp -I$R/include -I$R/tests/native $L -o /tmp/shim_cs || exit 1
{
echo "## concurrent_evals: 4 threads x 4 scripts x {HBM, pinned host, XTC file, XTC in HBM}"
for a in "300 48 6000" "120 96 30000" "40 64 120000"; do timeout 100 /tmp/concurrent_evals $a /tmp 2>&1 | grep -v amdgpu.ids | tail -2; echo "rc=$? ($a)"; done
echo "## stress_readahead with the side-by-side filtered evaluation (every third iteration)"
for s in 21 22 23; do timeout 120 /tmp/stress_ra 300 240 30000 $s 2>&1 | grep -v amdgpu.ids | tail -2; echo "rc=$? (seed $s)"; done
timeout 120 /tmp/stress_ra 120 240 30000 24 sdf 2>&1 | grep -v amdgpu.ids | tail -2; echo "rc=$? (sdf)"
echo "## shim_callsites (Eval Full and Eval Filt one after the other, then side by side), 40 runs of 96 frames"
ok=0; for i in $(seq 40); do timeout 30 /tmp/shim_cs 96 > /tmp/shim.out 2>&1 && ok=$((ok+1)) || { tail -3 /tmp/shim.out; }; done; echo "$ok of 40 runs OK; last: $(tail -1 /tmp/shim.out)"
} > $O/campaign.txt 2>&1
cat $O/campaign.txt
