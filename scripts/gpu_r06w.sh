#!/bin/bash
# round 6 campaign on the MI355X (final tree): random RDF / SDF / distance scenarios with the round's switches drawn in (computed selection index,
# small selections through the buckets, capacity sampling), pool patterns incl. the deferred-settle mode, the C++ read-ahead stress, and the
# reference's own call sites (oracle/_ref/ref_callsites) at several trajectory lengths.  usage: gpu_r06w.sh [tag] [seed0]
TAG=${1:-r06w}; S0=${2:-31}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "Round 6 campaign on the MI355X, final tree (scripts/gpu_r06w.sh $TAG $S0):"
echo "== fuzz_gpu.py: RDF + SDF + distance scenarios vs the oracle"
timeout 2400 python scripts/fuzz_gpu.py $((S0 * 100)) $((S0 * 100 + ${FUZZ_N:-240})) 20 2>&1 | tail -3
echo "== fuzz_pool.py: pool patterns (plain, then deferred settle)"
timeout 1800 python scripts/fuzz_pool.py $((S0 * 1000)) $((S0 * 1000 + ${POOL_N:-1200})) gpu 2>&1 | tail -2
timeout 1800 python scripts/fuzz_pool.py $((S0 * 1000 + 50000)) $((S0 * 1000 + 50000 + ${POOL_N:-1200} / 2)) gpu lone 2>&1 | tail -2
echo "== stress_readahead.cpp"
g++ -std=c++17 -O2 tests/native/stress_readahead.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/stress_ra
for k in $(seq 0 $((${STRESS_N:-4} - 1))); do timeout 600 /tmp/stress_ra 600 240 30000 $((S0 + k)) 2>&1 | tail -1; done
for k in 104 105; do timeout 600 /tmp/stress_ra 200 240 30000 $((S0 + k)) sdf 2>&1 | tail -1; done
echo "== the reference's own call sites through the shim (oracle/_ref/ref_callsites F tmpdir)"
for F in 16 33 64 100 257; do mkdir -p /tmp/rc$F; timeout 600 oracle/_ref/ref_callsites $F /tmp/rc$F 2>/dev/null | tail -1; done
echo "== shim_default_script / shim_callsites at other lengths"
for F in 20 77; do timeout 300 tests/native/shim_default_script $F 2>/dev/null | tail -1; timeout 300 tests/native/shim_callsites $F 2>/dev/null | tail -1; done
} > $OUT/campaign.txt 2>&1
cat $OUT/campaign.txt | cut -c1-220
