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
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") not in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_SALU"):
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            per[(k, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"] or 0)
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
res = json.load(open(out)) if os.path.exists(out) else {}
# the kernels these counters were collected on: bench.py compares this with the source it runs and says so when they differ
import hashlib
_ksrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "viamd_amd", "csrc", "vmd_kernels.hip")
_ksha = hashlib.sha256(open(_ksrc, "rb").read()).hexdigest()[:16] if os.path.exists(_ksrc) else None
res[workload] = {"frames_per_launch": frames_per_launch, "source": os.path.basename(src.rstrip("/")), "kernels_sha256_16": _ksha, "kernels": {}}
for k in sorted(acc):
    def full(vals):
        # a kernel may also be launched on a few frames only (k_cells_bin in counting mode): keep the full-batch dispatches
        m = max(vals)
        big = [v for v in vals if v >= 0.5 * m] or vals
        return big
    f = full(acc[k].get("FETCH_SIZE", [0])); w = full(acc[k].get("WRITE_SIZE", [0]))
    fetch, write = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
    res[workload]["kernels"][k] = {"fetch_bytes_per_launch_raw": fetch, "write_bytes_per_launch": write,
                                   "hbm_bytes_per_launch_raw": fetch + write,
                                   "hbm_bytes_per_launch_read_x2": 2 * fetch + write,
                                   "hbm_bytes_per_frame_raw": (fetch + write) / frames_per_launch}
    for cname, key in (("SQ_INSTS_VALU", "valu_insts_per_launch"), ("SQ_INSTS_SALU", "salu_insts_per_launch")):
        v = acc[k].get(cname)
        if v:
            v = full(v)
            res[workload]["kernels"][k][key] = sum(v) / len(v)     # wave-level instructions
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res[workload]["kernels"].get("k_rdf_pencil", {}), indent=1))
