"""oracle/_ref: the compilable part of the reference for this path, built from the sources WHERE THEY LIE.

    python oracle/make_ref.py        # /root/reference/src/{main.cpp,viamd.h,task_system.h} -> oracle/_ref/*.inc -> oracle/_ref/libviamd_ref.so
                                     #                                                       + oracle/_ref/ref_callsites (a host program)

TEST INFRASTRUCTURE.  Two products, both git-ignored, both generated from /root/reference and never committed:

1. libviamd_ref.so (round 4): the consumer-side histogram functions (free_histogram, compute_histogram, compute_histogram_masked,
   downsample_histogram, scale_histogram: src/main.cpp:132-261), cut out of the reference by their signatures - the line numbers are
   checked, not trusted - and compiled with oracle/ref_stubs.cpp around them.  tests/test_ref_pin.py holds oracle and product against it.

2. ref_callsites (round 6): VIAMD's OWN evaluation call sites as a host program of the drop-in boundary.  Cut
   verbatim, by signature / anchor line:
       src/viamd.h      DISPLAY_PROPERTY_MAX_* (:67-69), struct DisplayProperty (:272-370)
       src/task_system.h   namespace task_system (the declarations; the test double behind them is tests/native/viamd_host_double.h)
       src/main.cpp     MEASURE_EVALUATION_TIME (:70), PROPERTY_COLORS (:123), the histogram functions (:132-261),
                        display_property_copy_param_from_old (:1236), init_display_properties (:1259), update_display_properties (:1502),
                        export_xvg (:5640), export_csv (:5685), export_cube (:5718), sample_range (:5834)   [viamd_export_slices.inc: no ImGui
                        types inside, so tests/native/shim_callsites.cpp includes it too - in place of the copy it used to re-type]
                        and the evaluation block of the main loop (:950-1040: eval_init -> free / create x 2 -> init_display_properties ->
                        fingerprint check -> clear_data -> pool task calling md_script_eval_frame_range, full and filtered)
   into oracle/_ref/*.inc; tests/native/ref_callsites.cpp includes them between tests/native/md_mock.h (the test double of mdlib's
   declarations), include/vmd_md_script_shim.h (the boundary under test) and tests/native/viamd_host_double.h (md_file_* -> stdio,
   VIAMD_LOG_*, the fields of ApplicationState the slices touch, a thread pool behind task_system).  ImGui / ImPlot TYPES come from
   the reference's vendored headers where they lie (-I /root/reference/ext/imgui, .../implot).  So the REFERENCE's code - not a
   re-typing - calls md_script_eval_* / md_script_vis_eval_payload and reads prop_data through the shim, and its exporters write the
   files vmd_export_* must reproduce byte for byte.  Because the program needs /root/reference to compile, it is built HERE (against the
   product library for the GPU box, against the emulator build by the CPU suite) and travels prebuilt in oracle/_ref/.

The reference sources never enter the repository; on the GPU box (no /root/reference) the prebuilt files travel with the snapshot.
mdlib itself (rdf / sdf / distance arithmetic) is an empty submodule: nothing else of this path can be built."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_ROOT = "/root/reference"
REF_MAIN = REF_ROOT + "/src/main.cpp"
REF_VIAMD_H = REF_ROOT + "/src/viamd.h"
REF_TASK_H = REF_ROOT + "/src/task_system.h"
OUT_DIR = os.path.join(HERE, "_ref")
INC = os.path.join(OUT_DIR, "viamd_main_slices.inc")
LIB = os.path.join(OUT_DIR, "libviamd_ref.so")
STUBS = os.path.join(HERE, "ref_stubs.cpp")

FUNCTIONS = ["free_histogram", "compute_histogram", "compute_histogram_masked", "downsample_histogram", "scale_histogram"]

# ---- round 6: the call sites
INC_VIAMD_H = os.path.join(OUT_DIR, "viamd_h_slices.inc")
INC_TASK_H = os.path.join(OUT_DIR, "task_system_h_slices.inc")
INC_CALLSITES = os.path.join(OUT_DIR, "viamd_callsite_slices.inc")      # display properties: needs ImGui / ImPlot types (compiles only here)
INC_EXPORT = os.path.join(OUT_DIR, "viamd_export_slices.inc")           # the exporters: plain C++ (also included by tests/native/shim_callsites.cpp)
INC_EVAL_BLOCK = os.path.join(OUT_DIR, "viamd_eval_block.inc")
CALLSITES_SRC = os.path.join(ROOT, "tests", "native", "ref_callsites.cpp")
CALLSITES_DEPS = [CALLSITES_SRC] + [os.path.join(ROOT, "tests", "native", h) for h in ("md_mock.h", "md_mock_eval.h", "viamd_host_double.h")] + \
                 [os.path.join(ROOT, "include", h) for h in ("vmd_md_script_shim.h", "vmd_eval.h")]
CALLSITES_EXE = os.path.join(OUT_DIR, "ref_callsites")            # linked against viamd_amd/libviamd_amd.so: what the GPU box runs
CALLSITE_FUNCTIONS = [r"static void display_property_copy_param_from_old\(", r"static void init_display_properties\(",
                      r"static void update_display_properties\("]
EXPORT_FUNCTIONS = [r"static bool export_xvg\(", r"static bool export_csv\(", r"static bool export_cube\(", r"static md_array\(float\) sample_range\("]


def _one(lines, pattern, what, path):
    pat = re.compile(pattern)
    hits = [i for i, l in enumerate(lines) if pat.match(l)]
    if len(hits) != 1:
        raise RuntimeError(f"{path}: expected exactly one line matching {what or pattern!r}, found {len(hits)}")
    return hits[0]


def _slice(lines, name):
    """the definition `static void <name>(...) {` up to its closing brace at column 0"""
    b = _one(lines, r"^static void %s\(" % re.escape(name), f"definition of {name}", REF_MAIN)
    return _to_closing(lines, b, "}")


def _to_closing(lines, b, closer):
    e = b
    while lines[e].rstrip("\n") != closer:
        e += 1
    return b, e + 1


def available():
    return os.path.exists(REF_MAIN)


def _emit(out, path, lines, b, e):
    out.write(f"// ---- {path}:{b + 1}-{e}\n")
    out.writelines(lines[b:e])
    out.write("\n")


def write_callsite_slices():
    """the round-6 slices -> oracle/_ref/*.inc (verbatim line ranges; the anchors are checked to be unique)"""
    os.makedirs(OUT_DIR, exist_ok=True)
    banner = "// GENERATED by oracle/make_ref.py from %s - verbatim line ranges, not part of the repository\n"
    vh = open(REF_VIAMD_H).readlines()
    with open(INC_VIAMD_H, "w") as out:
        out.write(banner % REF_VIAMD_H)
        for i, l in enumerate(vh):
            if l.startswith("#define DISPLAY_PROPERTY_MAX_"):
                _emit(out, REF_VIAMD_H, vh, i, i + 1)
        b, e = _to_closing(vh, _one(vh, r"^struct DisplayProperty \{", None, REF_VIAMD_H), "};")
        _emit(out, REF_VIAMD_H, vh, b, e)
    th = open(REF_TASK_H).readlines()
    with open(INC_TASK_H, "w") as out:
        out.write(banner % REF_TASK_H)
        b = _one(th, r"^namespace task_system \{", None, REF_TASK_H)
        e = _one(th, r"^\}\s*// namespace task_system", None, REF_TASK_H) + 1
        _emit(out, REF_TASK_H, th, b, e)
    mc = open(REF_MAIN).readlines()
    with open(INC_CALLSITES, "w") as out:
        out.write(banner % REF_MAIN)
        for pattern in (r"^#define MEASURE_EVALUATION_TIME ", r"^constexpr uint32_t PROPERTY_COLORS\[\]"):
            i = _one(mc, pattern, None, REF_MAIN)
            _emit(out, REF_MAIN, mc, i, i + 1)
        for name in FUNCTIONS:
            _emit(out, REF_MAIN, mc, *_slice(mc, name))
        for pattern in CALLSITE_FUNCTIONS:
            _emit(out, REF_MAIN, mc, *_to_closing(mc, _one(mc, "^" + pattern + r".*\{\s*$", None, REF_MAIN), "}"))     # the definition, not a forward declaration
    with open(INC_EXPORT, "w") as out:
        out.write(banner % REF_MAIN)
        for pattern in EXPORT_FUNCTIONS:
            _emit(out, REF_MAIN, mc, *_to_closing(mc, _one(mc, "^" + pattern + r".*\{\s*$", None, REF_MAIN), "}"))
    with open(INC_EVAL_BLOCK, "w") as out:
        out.write(banner % REF_MAIN)
        # the block `if (num_frames > 0) { if (state.script.eval_init) ... "Eval Full" ... "Eval Filt" ... }` of the main loop: from the line
        # above the one mention of `if (state.script.eval_init) {` to the line before `recenter_update(&state);`
        b = _one(mc, r"^\s+if \(state\.script\.eval_init\) \{", None, REF_MAIN) - 1
        if not re.match(r"^\s+if \(num_frames > 0\) \{", mc[b]):
            raise RuntimeError(f"{REF_MAIN}:{b + 1}: expected `if (num_frames > 0) {{` above the eval_init block")
        e = _one(mc, r"^\s+recenter_update\(&state\);", None, REF_MAIN)
        while mc[e - 1].strip() == "":
            e -= 1
        _emit(out, REF_MAIN, mc, b, e)
    return [INC_VIAMD_H, INC_TASK_H, INC_CALLSITES, INC_EXPORT, INC_EVAL_BLOCK]


def callsites_compile_cmd(exe, link_args, opt="-O2"):
    """g++ command line of tests/native/ref_callsites.cpp (needs /root/reference: ImGui / ImPlot headers and the slices)"""
    return ["g++", "-std=c++20", opt, "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-format", "-Wno-format-security", "-Wno-type-limits",
            "-ffp-contract=off", "-fno-fast-math",
            # src/main.cpp:1314-1315 passes sizeof(item.unit_str) (64) as the capacity of item.unit_str[k] (32 bytes): harmless for the short
            # unit strings, but Ubuntu's default _FORTIFY_SOURCE aborts on the declared bound alone - the reference is compiled as it is
            "-U_FORTIFY_SOURCE", "-D_FORTIFY_SOURCE=0",
            CALLSITES_SRC, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "native"), "-I" + HERE,
            "-I" + REF_ROOT + "/ext/imgui", "-I" + REF_ROOT + "/ext/implot"] + list(link_args) + ["-lpthread", "-o", exe]


def slices_available():
    """the round-6 slices exist (generated here, or travelled to the GPU box in oracle/_ref/)"""
    if available():
        deps = [REF_MAIN, REF_VIAMD_H, REF_TASK_H, os.path.abspath(__file__)]
        incs = [INC_VIAMD_H, INC_TASK_H, INC_CALLSITES, INC_EXPORT, INC_EVAL_BLOCK]
        if not all(os.path.exists(i) for i in incs) or min(os.path.getmtime(i) for i in incs) < max(os.path.getmtime(d) for d in deps):
            write_callsite_slices()
        return True
    return os.path.exists(INC_EXPORT)


def build_callsites(force=False):
    """-> oracle/_ref/ref_callsites (VIAMD's own call sites against the PRODUCT library), or None when it can neither be built nor found.
    The product library must exist (viamd_amd.build.build()); callers on the CPU suite link the emulator build themselves."""
    if not available():
        return CALLSITES_EXE if os.path.exists(CALLSITES_EXE) else None
    lib = os.path.join(ROOT, "viamd_amd", "libviamd_amd.so")
    if not os.path.exists(lib):
        return None
    deps = [REF_MAIN, REF_VIAMD_H, REF_TASK_H, os.path.abspath(__file__), lib] + CALLSITES_DEPS
    if not force and os.path.exists(CALLSITES_EXE) and all(os.path.getmtime(CALLSITES_EXE) >= os.path.getmtime(d) for d in deps):
        return CALLSITES_EXE
    write_callsite_slices()
    subprocess.check_call(callsites_compile_cmd(CALLSITES_EXE, ["-L" + os.path.join(ROOT, "viamd_amd"), "-lviamd_amd", "-L/opt/rocm/lib",
                                                                "-Wl,-rpath,$ORIGIN/../../viamd_amd", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]))
    return CALLSITES_EXE


def build(force=False):
    """-> path of the library, or None when neither the reference nor a prebuilt library is there"""
    if not available():
        return LIB if os.path.exists(LIB) else None
    deps = [REF_MAIN, STUBS, os.path.abspath(__file__)]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    lines = open(REF_MAIN).readlines()
    with open(INC, "w") as out:
        out.write("// GENERATED by oracle/make_ref.py from %s - verbatim line ranges, not part of the repository\n" % REF_MAIN)
        for name in FUNCTIONS:
            b, e = _slice(lines, name)
            out.write(f"// ---- {REF_MAIN}:{b + 1}-{e}\n")
            out.writelines(lines[b:e])
            out.write("\n")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
                           "-I", HERE, STUBS, "-o", LIB])
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print(p or "no reference and no prebuilt library")
    print(build_callsites(force="--force" in sys.argv) or "ref_callsites: not built (no reference / no product library) and no prebuilt program")
