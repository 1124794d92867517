"""The drop-in boundary: the hipcc-built shared library loads and exports every symbol include/*.h declares.
No compute calls (this container has no GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("vmd_eval.h", "vmd_hip.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(vmd_[a-z0-9_]+)\s*\(", src))
    return names


@pytest.fixture(scope="module")
def product_lib():
    from viamd_amd import build
    build.build()                       # hipcc cross-compiles gfx950 without a GPU
    from viamd_amd import default_lib
    return default_lib()


def test_every_declared_symbol_is_exported(product_lib):
    decl = declared_symbols()
    assert len(decl) >= 45
    missing = [n for n in sorted(decl) if not hasattr(product_lib.cdll, n)]
    assert not missing, f"libviamd_amd.so lacks {missing}"
    from viamd_amd import _lib
    bound = {s[0] for s in _lib.SIGNATURES}
    assert decl == bound, f"ctypes table and headers disagree: {sorted(decl ^ bound)}"


def test_library_is_gfx950_code_object(product_lib):
    data = open(product_lib.path, "rb").read()
    assert b"gfx950" in data and b"k_rdf_pencil" in data


def test_no_device_fails_loudly(product_lib):
    """There is no CPU path: without a GPU eval creation must fail with a message, not fall back."""
    import viamd_amd as V
    if product_lib.vmd_device_count() > 0:
        pytest.skip("a GPU is visible here")
    ir = V.ScriptIR(product_lib)
    ir.add_rdf("g", [0, 1], [0, 1], 5.0)
    assert ir.property_names() == ["g"] and ir.property_flags("g") == V.FLAG_DISTRIBUTION and ir.valid()
    with pytest.raises(V.VmdError, match="no usable HIP device"):
        V.ScriptEval(4, ir)
    with pytest.raises(V.VmdError):
        V.DeviceTrajectory(2, 10, lib=product_lib)


def test_ir_validation(product_lib):
    import viamd_amd as V
    ir = V.ScriptIR(product_lib)
    ir.add_rdf("a", [0], [1], 5.0)
    with pytest.raises(V.VmdError, match="already defined"):
        ir.add_rdf("a", [0], [1], 5.0)
    with pytest.raises(V.VmdError):
        ir.add_rdf("b", [], [1], 5.0)
    with pytest.raises(V.VmdError):
        ir.add_rdf("c", [0], [1], (4.0, 2.0))
    with pytest.raises(V.VmdError):
        ir.add_sdf("d", [[0, 1]], [2], -1.0)
    ir.add_sdf("v", [[0, 1], [2, 3]], [4, 5], 8.0)
    ir.add_distance("d1", [0], [1])
    assert ir.property_count() == 3
    assert [ir.property_flags(n) for n in ir.property_names()] == [V.FLAG_DISTRIBUTION, V.FLAG_VOLUME, V.FLAG_TEMPORAL]
    fp = ir.fingerprint()
    ir.add_distance("d2", [0], [2], V.DIST_MIN)
    assert ir.fingerprint() != fp


def test_log_hook_receives_failures(emu_lib, capfd):
    """md_log_register analogue (VIAMD shows mdlib's messages as toasts, src/main.cpp:384-420): a registered callback gets the
    evaluator's failure messages instead of stderr; NULL restores stderr."""
    from viamd_amd import _lib as L
    seen = []
    cb = L.LOG_FN(lambda level, msg, user: seen.append((level, msg.decode())))
    emu_lib.vmd_log_register(cb, None)
    try:
        assert not emu_lib.vmd_eval_frame_range(None, None, None, None, 0, 1)
    finally:
        emu_lib.vmd_log_register(L.LOG_FN(0), None)
    assert seen and seen[0][0] == 2 and "NULL argument" in seen[0][1]
    assert "NULL argument" in emu_lib.last_error()
    assert "NULL argument" not in capfd.readouterr().err
    assert not emu_lib.vmd_eval_frame_range(None, None, None, None, 0, 1)
    assert "NULL argument" in capfd.readouterr().err
