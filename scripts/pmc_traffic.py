"""Turn a scripts/gpu_pmc.sh output directory into profiles/pmc_traffic.json (HBM-side traffic per kernel launch).

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3), collected in separate --pmc passes (MI355X_MICROARCH.md, rocprofv3 PMC slots).
gfx950 note from the same guide: FETCH_SIZE under-reports wide coalesced streaming reads by exactly 2x; other access widths are
uncalibrated, so both the raw value and the 2x-corrected read figure are stored."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, workload, frames_per_launch, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
# steps the profiled bench command ran (warm-up + timed; scripts/gpu_pmc.sh: --steps 1 --warmup 1) and frames it evaluates per step
steps_in_run = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
frames_per_step = float(sys.argv[6]) if len(sys.argv) > 6 else frames_per_launch
COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")
acc = defaultdict(lambda: defaultdict(list))
passes = defaultdict(lambda: defaultdict(int))           # kernel -> counter -> number of passes that collected it (SQ_INSTS_VALU sits in two)
marked = 0
for f in sorted(glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(float)
    rows = list(csv.DictReader(open(f)))
    # round 6: bench.py dispatches k_marker_timed_region where its timed region begins - what comes before (the warm-up step: capacity
    # sampling, overflow repeats, first touches) is not the steady state and is dropped; steps_in_run then counts the TIMED steps only
    marks = [int(r["Dispatch_Id"]) for r in rows if r.get("Kernel_Name", "").startswith("k_marker_timed_region")]
    first_kept = max(marks) if marks else -1
    marked += 1 if marks else 0
    for row in rows:
            if row.get("Counter_Name") not in COUNTERS or int(row["Dispatch_Id"]) <= first_kept:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            per[(k, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"] or 0)
    seen = set()
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
        seen.add((k, c))
    for k, c in seen:
        passes[k][c] += 1
res = json.load(open(out)) if os.path.exists(out) else {}
# the kernels these counters were collected on: bench.py compares this with the source it runs and says so when they differ
import hashlib
_ksrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "viamd_amd", "csrc", "vmd_kernels.hip")
_ksha = hashlib.sha256(open(_ksrc, "rb").read()).hexdigest()[:16] if os.path.exists(_ksrc) else None
res[workload] = {"frames_per_launch": frames_per_launch, "frames_per_step": frames_per_step, "steps_in_run": steps_in_run,
                 "timed_region_only": bool(marked),        # True: dispatches before bench.py's k_marker_timed_region were dropped, steps_in_run = timed steps
                 "source": os.path.basename(src.rstrip("/")), "kernels_sha256_16": _ksha, "kernels": {}}
for k in sorted(acc):
    def full(vals):
        # a kernel may also be launched on a few frames only (k_cells_bin in counting mode): keep the full-batch dispatches
        m = max(vals)
        big = [v for v in vals if v >= 0.5 * m] or vals
        return big
    def per_step(cname):
        # every dispatch of the kernel (all template variants, all passes of a multi-pass script), summed over one step of the bench
        v = acc[k].get(cname)
        return sum(v) / max(1, passes[k][cname]) / steps_in_run if v else None
    f = full(acc[k].get("FETCH_SIZE", [0])); w = full(acc[k].get("WRITE_SIZE", [0]))
    fetch, write = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
    e = {"fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
         "hbm_bytes_per_launch_raw": fetch + write,
         "hbm_bytes_per_launch_read_x2": 2 * fetch + write,
         "hbm_bytes_per_frame_raw": (fetch + write) / frames_per_launch,
         "dispatches_per_step": len(acc[k].get("FETCH_SIZE", [])) / steps_in_run}
    fs, ws = per_step("FETCH_SIZE"), per_step("WRITE_SIZE")
    if fs is not None and ws is not None:
        e["hbm_bytes_per_step_read_x2"] = (2 * fs + ws) * 1024          # the guide's gfx950 correction: FETCH_SIZE counts half a streaming read
        e["hbm_bytes_per_step_raw"] = (fs + ws) * 1024
    for cname, key in (("SQ_INSTS_VALU", "valu_insts_per_launch"), ("SQ_INSTS_SALU", "salu_insts_per_launch")):
        v = acc[k].get(cname)
        if v:
            v = full(v)
            e[key] = sum(v) / len(v)     # wave-level instructions
    for cname, key in (("SQ_INSTS_VALU", "valu_insts_per_step"), ("SQ_INSTS_SALU", "salu_insts_per_step"),
                       ("SQ_ACTIVE_INST_VALU", "active_inst_valu_per_step"), ("GRBM_GUI_ACTIVE", "gui_active_per_step")):
        v = per_step(cname)
        if v is not None:
            e[key] = v
    if e.get("active_inst_valu_per_step") and e.get("gui_active_per_step"):
        # SQ_ACTIVE_INST_VALU: quad-cycles a SIMD's vector ALU is busy, summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE: cycles the kernel
        # is resident, summed over the 8 XCDs (128 SIMDs each).  busy = share of the SIMD cycles with the VALU at work: a counter ratio,
        # no calibrated peak in it, cannot exceed 1
        e["valu_busy"] = 4.0 * e["active_inst_valu_per_step"] / (128.0 * e["gui_active_per_step"])
    res[workload]["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res[workload]["kernels"].get("k_rdf_pencil", {}), indent=1))
