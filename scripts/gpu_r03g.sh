#!/bin/bash
# Round 3, mid-round checkpoint: full GPU test suite, the default bench line (c3 + secondary + CPU baselines), XTC paths with the final
# batch policy, and the 2-ranks-on-one-GPU RCCL experiment.
TAG=${1:-r03g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket"; } > $OUT/device.txt
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
echo "== bench default"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
cb = d.get("cpu_baseline", {})
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step", {k: round(v, 1) for k, v in d["kernel_ms"].items()})
print("   cpu: scalar", round(cb.get("scalar", {}).get("value", 0), 2), "simd", round((cb.get("simd") or {}).get("value", 0), 2), "cores", cb.get("cores"), "node", cb.get("extrapolated_to_node"), "gpu/cpu", round(d.get("gpu_over_cpu", 0)), "gpu/cpu(node)", d.get("gpu_over_cpu_node"))
print("   columns", d.get("pair_kernel_columns"))
for k, v in d.get("secondary", {}).items():
    print(k, round(v["value"]), "frames/s", round(v["ms_per_step"], 3), "ms/step", {a: round(b, 2) for a, b in v["kernel_ms"].items()}, "kernel frac", round(v["roofline"]["frac"], 4), "step frac", round(v["roofline"]["step_level"]["frac"], 4))
PY
tail -2 $OUT/bench_default.err | grep -v amdgpu.ids
run() {  # name, bench args...
  n=$1; shift
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --steps 5 "$@" > $OUT/bench_c2_$n.json 2>> $OUT/bench_xtc.err
  python -c "import json;d=json.load(open('$OUT/bench_c2_$n.json'));print('$n', round(d['value']), 'frames/s', d['config'].get('frames_decompressed_on_device_per_step'), {k: round(v, 1) for k, v in d.get('kernel_ms', {}).items()})"
}
echo "== c2 from XTC, final batch policy"
run xtc_file --traj xtc
run xtc_file_rw --traj xtc --rigid-water
run xtc_host32 --traj xtc --opt xtc_device_decode=0 --opt load_threads=32
run xtc_resident --traj xtc-resident
echo "== two RCCL ranks on ONE GPU (does RCCL accept duplicate devices?)"
python - <<PY > $OUT/build_demo.log 2>&1
import sys; sys.path.insert(0, "tests")
import test_native; print(test_native.build_reduce_demo())
PY
rm -f /tmp/rccl2.id
( for r in 0 1; do HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 120 tests/native/cabi_reduce_demo 2 $r /tmp/rccl2.id 24 > $OUT/reduce2_rank$r.log 2>&1 & done; wait )
for r in 0 1; do echo "rank $r:"; tail -3 $OUT/reduce2_rank$r.log; done
echo done
