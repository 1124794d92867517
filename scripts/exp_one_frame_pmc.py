"""where do the 170 us of a ONE-frame pair launch go?  Counters, not hypotheses.
Run under rocprofv3 --pmc ... --kernel-trace (scripts/gpu_r05b.sh): c2's O-O RDF, one warm-up evaluation of all 64 frames, then 32
one-frame calls, 8 four-frame calls and one 64-frame call - k_rdf_pencil dispatches 1..32, 33..40 and 41 in the trace.
`python scripts/exp_one_frame_pmc.py summarize <dir>` turns the passes into the table (duration from the kernel trace, counters per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GROUPS = (("warm-up, 64 frames", 0, 1), ("1 frame per launch", 1, 33), ("4 frames per launch", 33, 41), ("64 frames per launch", 41, 42))


def run():
    import viamd_amd as V
    from viamd_amd import script, synth
    lib = V.default_lib()
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        lib.vmd_set_option(k.encode(), int(v))
    N, box, F = 100002, 100.0, 64
    traj = synth.make_device_trajectory(V, 2, N, box, F, 0)
    topo = synth.water_box_topology(N, 0)
    sysm = V.MolSystem(N, mass=topo.mass, unitcell=V.make_unitcell(box))
    ir, _ = script.compile_script("g = rdf(element('O'), element('O'), 12.0);", topo)
    ev = V.ScriptEval(F, ir)
    assert ev.frame_range(sysm, traj, 0, F)
    ev.clear_data()
    for f in range(32):
        assert ev.frame_range(sysm, traj, f, f + 1)
    for f in range(32, 64, 4):
        assert ev.frame_range(sysm, traj, f, f + 4)
    ev.clear_data()
    assert ev.frame_range(sysm, traj, 0, F)
    ev.close()


def summarize(src):
    dur = {}                                         # dispatch id -> ns (any pass: the kernel trace rides along with every pass)
    per = defaultdict(lambda: defaultdict(dict))     # pass dir -> dispatch id -> counter -> value
    for d in sorted(glob.glob(os.path.join(src, "p*"))):
        if not os.path.isdir(d):
            continue
        ids = []
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_rdf_pencil" in row["Kernel_Name"]:
                    ids.append((int(row["Dispatch_Id"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        ids.sort()
        order = {did: i for i, (did, _) in enumerate(ids)}
        for did, ns in ids:
            dur.setdefault((os.path.basename(d), order[did]), ns)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(float)
            for row in csv.DictReader(open(f)):
                if "k_rdf_pencil" in row["Kernel_Name"]:
                    acc[(int(row["Dispatch_Id"]), row["Counter_Name"])] += float(row["Counter_Value"] or 0)
            for (did, c), v in acc.items():
                if did in order:
                    per[os.path.basename(d)][order[did]][c] = v
    counters = sorted({c for p in per.values() for dd in p.values() for c in dd})
    print(f"# k_rdf_pencil, c2 O-O RDF (33 334 atoms per frame); mean per dispatch of each group; duration from the kernel trace of the same pass")
    print(f"{'group':24s} {'pass':5s} {'n':>3s} {'dur us':>9s} " + " ".join(f"{c:>20s}" for c in counters))
    rows = {}
    for p in sorted(per):
        for name, a, b in GROUPS:
            ids = [i for i in range(a, b) if i in per[p]]
            if not ids:
                continue
            du = sum(dur.get((p, i), 0) for i in ids) / len(ids) / 1e3
            vals = {c: sum(per[p][i].get(c, 0.0) for i in ids) / len(ids) for c in counters if any(c in per[p][i] for i in ids)}
            rows[(name, p)] = (du, vals)
            print(f"{name:24s} {p:5s} {len(ids):3d} {du:9.1f} " + " ".join(f"{vals[c]:20.6g}" if c in vals else f"{'':>20s}" for c in counters))
    print("# derived (GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_* over all SEs):")
    for (name, p), (du, v) in rows.items():
        out = []
        if "GRBM_GUI_ACTIVE" in v and du > 0:
            out.append(f"clock while the kernel is resident = GRBM_GUI_ACTIVE / 8 / duration = {v['GRBM_GUI_ACTIVE'] / 8 / (du * 1e3):.3f} GHz")
        if "SQ_WAVES" in v:
            out.append(f"waves {v['SQ_WAVES']:.0f}")
        if "SQ_WAVE_CYCLES" in v and v.get("SQ_WAVES"):
            out.append(f"mean wave lifetime {4 * v['SQ_WAVE_CYCLES'] / v['SQ_WAVES']:.0f} cycles")
        if "SQ_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            out.append(f"SQ busy / GUI active = {v['SQ_BUSY_CYCLES'] / v['GRBM_GUI_ACTIVE']:.2f}")
        if "SQ_INSTS_VALU" in v and v.get("SQ_WAVES"):
            out.append(f"VALU insts per wave {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.0f}")
        if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
            out.append(f"waiting {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.2f} of wave cycles")
        if "SQ_ACTIVE_INST_VALU" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
            out.append(f"VALU active {v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES']:.2f} of wave cycles")
        if out:
            print(f"{name:24s} {p:5s} " + "; ".join(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "summarize":
        summarize(sys.argv[2])
    else:
        run()
