#!/bin/bash
# Round 3: the evaluator life-cycle stress run: 2 x 1000 life cycles with HSA_ENABLE_SDMA=1 / 0 and
# AMD_LOG_LEVEL=1; the GPU tests added since the checkpoint (bonds, spec switches, stress); the XTC file path with the final policy.
TAG=${1:-r03h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu (late tests + native)"
timeout 900 python -m pytest tests/test_zz_late_gpu.py tests/test_native.py tests/test_xdr.py -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_new.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest_gpu_new.log
echo "== stress"
g++ -std=c++17 -O2 tests/native/stress_eval.cpp -Iinclude viamd_amd/libviamd_amd.so -Wl,-rpath,$R/viamd_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread -o /tmp/stress_eval || exit 1
for sdma in 1 0; do
  t0=$(date +%s.%N)
  HSA_ENABLE_SDMA=$sdma AMD_LOG_LEVEL=1 timeout 900 /tmp/stress_eval 1000 48 > $OUT/stress_sdma$sdma.out 2> $OUT/stress_sdma$sdma.err; rc=$?
  t1=$(date +%s.%N)
  echo "HSA_ENABLE_SDMA=$sdma rc=$rc $(tail -1 $OUT/stress_sdma$sdma.out) wall=$(echo "$t1 - $t0" | bc) s, stderr lines: $(wc -l < $OUT/stress_sdma$sdma.err)"
  tail -3 $OUT/stress_sdma$sdma.err
done
run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 "$@" > $OUT/bench_c2_$n.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$n.json'));print('$n', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), {k: round(v, 1) for k, v in d.get('kernel_ms', {}).items()})"
}
echo "== c2 from XTC, final policy"
run xtc_file --traj xtc
run xtc_file_rw --traj xtc --rigid-water
echo done
