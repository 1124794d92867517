"""A C++ host (no Python, no torch) driving the C ABI the way VIAMD drives mdlib: pool threads on one eval, GUI-thread
polling, interrupt + restart (tests/native/cabi_demo.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "cabi_demo.cpp")
EXE = os.path.join(ROOT, "tests", "native", "cabi_demo")


def build_demo():
    from viamd_amd import build
    lib = build.build()
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(SRC), os.path.getmtime(lib)):
        return EXE
    subprocess.check_call(["g++", "-std=c++17", "-O2", SRC, "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "viamd_amd"),
                           "-lviamd_amd", "-L/opt/rocm/lib", "-Wl,-rpath,$ORIGIN/../../viamd_amd", "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-o", EXE])
    return EXE


def test_native_host_builds_against_the_c_abi():
    exe = build_demo()
    assert os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_native_host_runs_like_viamd():
    exe = build_demo()
    out = subprocess.run([exe, "96"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK frames=96"), out.stdout
