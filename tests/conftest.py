import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement (test infrastructure, parity unpinned — oracle/SPEC.md)."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libviamd_emu.so")
from viamd_amd import build as _vb          # the ONE list of product sources (viamd_amd/build.py): the emulator compiles exactly those
EMU_SOURCES = list(_vb.SOURCES) + [os.path.join(EMU_DIR, "emu.cpp")]
EMU_DEPS = EMU_SOURCES + [os.path.join(EMU_DIR, "hip", "hip_runtime.h")] + list(_vb.HEADERS)


def build_emu():
    """g++ build of the *same* kernel + host sources against the test-only SIMT emulator (tests/emu).
    VIAMD_EMU_SANITIZE=address,undefined (or thread) builds an instrumented copy next to it: run the suite under it with
    LD_PRELOAD=$(g++ -print-file-name=libasan.so) (scripts/sanitize_emu.sh)."""
    san = os.environ.get("VIAMD_EMU_SANITIZE", "")
    out = EMU_LIB if not san else os.path.join("/tmp", "libviamd_emu_" + san.replace(",", "_") + ".so")   # never into the tree: in-tree .so files travel to the GPU box
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in EMU_DEPS):
        return out
    # several pytest-xdist workers may find the library missing at once: each builds into its own files and renames the library into place
    from concurrent.futures import ThreadPoolExecutor
    tmp = f"{out}.{os.getpid()}.tmp"
    objdir = os.path.join("/tmp", f"viamd_emu_obj_{san.replace(',', '_') or 'plain'}_{os.getpid()}")
    os.makedirs(objdir, exist_ok=True)
    cflags = ["-O2", "-g", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mavx2", "-mfma", "-I" + EMU_DIR, "-I" + os.path.join(ROOT, "include")]
    if san:
        cflags = ["-fsanitize=" + san, "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"] + cflags

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        subprocess.check_call(["g++"] + cflags + ["-x", "c++", "-c", src, "-o", obj])
        return obj

    try:
        with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_one, EMU_SOURCES))
        subprocess.check_call(["g++", "-shared"] + (["-fsanitize=" + san] if san else []) + objs + ["-ldl", "-o", tmp])
        os.replace(tmp, out)
    finally:
        import shutil
        shutil.rmtree(objdir, ignore_errors=True)
        if os.path.exists(tmp):
            os.remove(tmp)
    return out


TWIN_LIB = os.path.join(ROOT, "tests", "native", "libviamd_amd_twin.so")


def build_twin():
    """hipcc build of the product sources with -DVMD_NO_INLINE_ASM: the C++ twins the emulator runs, compiled for gfx950.
    The GPU suite evaluates the same inputs through this library and through the product and compares the accumulators bit for
    bit (tests/test_zz_late_gpu.py) - that ties the inline-asm blocks to the code the CPU suite covers.  Test-only: it lives
    under tests/native, never next to the package, and nothing in viamd_amd/ loads it."""
    from viamd_amd import build as vb
    deps = vb.SOURCES + vb.HEADERS
    if os.path.exists(TWIN_LIB) and all(os.path.getmtime(TWIN_LIB) >= os.path.getmtime(s) for s in deps):
        return TWIN_LIB
    return vb.build(force=False, extra_flags=("-DVMD_NO_INLINE_ASM",), out=TWIN_LIB)


@pytest.fixture(scope="session")
def emu_lib():
    """Kernels + host code compiled for the CPU SIMT emulator: logic checks only, never the product path."""
    from viamd_amd import VmdLib
    return VmdLib(build_emu())


@pytest.fixture(scope="session")
def gpu_lib():
    """The product: hipcc-built libviamd_amd.so on a real GPU.  Fails (not skips) when the library or GPU is missing."""
    # torch first, as in bench.py: a few GPU tests take device memory from torch, and its bundled HIP runtime only finds the GPU
    # when it is loaded before the library pulls in the system one (the other order: "No HIP GPUs are available", r03a)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    from viamd_amd import default_lib
    lib = default_lib()
    assert lib.vmd_device_count() > 0, "no HIP device visible: -m gpu tests must run on the MI355X box"
    return lib


@pytest.fixture(scope="session", params=["emu", "product"])
def host_lib(request):
    """For entry points that are pure host code (file readers / writers, script front-end, post-processing): the g++ emulator
    build and the hipcc-built product library (clang, -O3) - the latter loads without a GPU, only device calls need one."""
    from viamd_amd import VmdLib, default_lib
    return VmdLib(build_emu()) if request.param == "emu" else default_lib()
