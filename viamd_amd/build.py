"""Build viamd_amd/libviamd_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
# the evaluator (round 6: one 4 300-line translation unit split by concern; shared structs in csrc/vmd_eval_internal.h)
EVAL_SOURCES = ["vmd_eval_runtime.cpp",      # errors, options, logging, profiling, the process-wide resource pool
                "vmd_eval_ir.cpp",           # property descriptors (vmd_ir_*)
                "vmd_eval_core.cpp",         # create / free / clear_data / interrupt, host views, accessors, sdf payload
                "vmd_eval_stage.cpp",        # static uploads, trajectory staging (device views, pinned batches, raw XTC frames + device decode)
                "vmd_eval_batch.cpp",        # grids, cell builds, batch planning, block reuse, process_range
                "vmd_eval_calls.cpp",        # how calls arrive: combining queue, read-ahead, deferred settle, vmd_eval_frame_range
                "vmd_eval_traj.cpp",         # trajectory kinds in HBM / pinned memory, decoder checkpoints, mapped-file windows
                "vmd_eval_post.cpp"]         # VIAMD's consumer-side histogram post-processing
SOURCES = [os.path.join(_HERE, "csrc", "vmd_kernels.hip"), os.path.join(_HERE, "csrc", "vmd_xtc_device.hip")] + \
          [os.path.join(_HERE, "csrc", f) for f in EVAL_SOURCES] + \
          [os.path.join(_HERE, "csrc", f) for f in ("vmd_dcd.cpp", "vmd_xdr.cpp", "vmd_script.cpp", "vmd_reduce.cpp", "vmd_export.cpp", "vmd_text.cpp")]
HEADERS = [os.path.join(ROOT, "include", "vmd_eval.h"), os.path.join(ROOT, "include", "vmd_hip.h"), os.path.join(_HERE, "csrc", "vmd_eval_internal.h")]
OUT = os.path.join(_HERE, "libviamd_amd.so")

# -ffp-contract=off: oracle/SPEC.md names every fused operation explicitly (fmaf); nothing else may be contracted.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value", "-Wno-pass-failed",
         "-Wno-inline-asm"]      # vmd_pop_hot0 names m0 (the base of ds_read_addtid_b32) as clobbered: "reserved register", by design


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X backend cannot be built (there is no CPU fallback)")
    return exe


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(s) <= t for s in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build(force=False, verbose=False, extra_flags=(), out=None):
    """one hipcc per source (in parallel, objects under build/), one link: a change to one file recompiles that file"""
    out = out or OUT
    if not force and out == OUT and up_to_date():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    tag = "obj" if not extra_flags else "obj_" + "_".join(f.lstrip("-D") for f in extra_flags)
    objdir = os.path.join(ROOT, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    newest_header = max(os.path.getmtime(h) for h in HEADERS + [os.path.abspath(__file__)])

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header):
            return obj
        cmd = [hipcc()] + cflags + ["-I", os.path.join(ROOT, "include"), "-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
