#!/bin/bash
# round 6, call A: the -m gpu suite on the new tree (the reference's own call sites among it), smoke, a read-only probe of the compute
# partition, a kernel + memcpy timeline of c4's 1 250-frame step (where do its 0.72 ms go?), the default line
T=${1:-r06a}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$T; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
{
  echo "== id"; id
  echo "== rocm-smi --showcomputepartition --showmemorypartition"; rocm-smi --showcomputepartition --showmemorypartition 2>&1
  echo "== sysfs partition files (mode, owner; writable by this user?)"
  for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition /sys/class/drm/card*/device/current_memory_partition; do
    [ -e "$f" ] && { ls -l "$f"; echo "   value: $(cat $f 2>&1)"; [ -w "$f" ] && echo "   WRITABLE by uid $(id -u)" || echo "   not writable by uid $(id -u)"; }
  done
  echo "== devices visible"; rocm-smi --showid 2>&1 | head -20; ls /dev/dri 2>&1; ls -l /dev/kfd 2>&1
  echo "== HIP_VISIBLE_DEVICES=$HIP_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES=$ROCR_VISIBLE_DEVICES"
} > $O/partition_probe.txt 2>&1
cat $O/partition_probe.txt | head -60
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_c4_1250 -o t -- python $R/bench.py --workload c4 --frames 1250 --steps 12 --warmup 4 --no-cpu-baseline --no-secondary > $O/trace_c4_1250.log 2>&1
echo "trace rc=$?"; tail -2 $O/trace_c4_1250.log | cut -c1-300
cd $R
python - <<PY
import csv, glob, os
O = "$O/trace_c4_1250"
rows = []
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:60], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
for f in glob.glob(O + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), "", ""))
rows.sort()
# the last 3 steps: everything after the third-last k_sdf_align... print the tail of the timeline relative to its start
tail = rows[-90:]
t0 = tail[0][0]
with open("$O/timeline_c4_1250.txt", "w") as out:
    for s, e, n, q, st in tail:
        out.write(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {n}  q{q} s{st}\n")
print(open("$O/timeline_c4_1250.txt").read()[-4500:])
PY
find $O/trace_c4_1250 -name "*.csv" -size +2M -delete
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("c3", round(d["value"]), "frames/s", round(d["ms_per_step"], 2), "ms/step frac", round(d["roofline"]["frac"], 4), "cpu", round(d["cpu_baseline"]["value"], 1), "gpu/cpu", round(d["gpu_over_cpu"], 1))
for k, v in d.get("secondary", {}).items(): print(k, round(v["value"]), round(v["ms_per_step"], 3), v.get("strong_scaling_bound_8_gpus"))
PY
du -sh $O
