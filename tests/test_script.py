"""Mini script front-end: BASELINE config strings -> property descriptors."""
import numpy as np
import pytest

import viamd_amd as V
from viamd_amd import script, synth


@pytest.fixture(scope="module")
def topo():
    return synth.water_box_topology(2000 + 3000, n_blob=2000)


def test_topology(topo):
    assert topo.num_atoms == 5000 and topo.num_residues == 200 + 1000
    assert topo.residue_name(0) == "ALA" and topo.residue_name(200) == "HOH"
    assert list(topo.residue_atoms(3)) == list(range(30, 40))
    assert topo.mass[2000] == np.float32(15.999) and topo.mass[2001] == np.float32(1.008)


def test_baseline_config_scripts_compile(emu_lib, topo):
    ir, info = script.compile_script("g = rdf(element('O') and water, element('O') and water, 12.0);", topo, lib=emu_lib)
    assert ir.property_names() == ["g"] and ir.property_flags("g") == V.FLAG_DISTRIBUTION
    np.testing.assert_array_equal(info["g"]["ref"], np.arange(2000, 5000, 3))
    ir, info = script.compile_script("""
        # default script shape of src/main.cpp:528
        s1 = resname("ALA")[2:8];
        r = rdf(element('C'), element('H'), 10.0);
        v = sdf(s1, element('H'), 10.0);
        d1 = distance(10, 30);
        d2 = distance_min(residue(3), water and element('O'));
        h = rdf(not element('H'), not element('H'), {2.0, 9.0});
    """, topo, lib=emu_lib)
    assert ir.property_names() == ["r", "v", "d1", "d2", "h"]
    assert info["v"]["structures"].shape == (7, 10) and info["v"]["structures"][0, 0] == 10      # residues 2..8, 1-based
    assert info["d1"]["a"].tolist() == [9] and info["d1"]["b"].tolist() == [29]                  # 1-based script indices
    assert info["d2"]["kind"] == "distance_min" and info["d2"]["a"].tolist() == list(range(20, 30))
    assert (info["h"]["rmin"], info["h"]["rmax"]) == (2.0, 9.0)
    assert [ir.property_flags(n) for n in ir.property_names()] == [V.FLAG_DISTRIBUTION, V.FLAG_VOLUME, V.FLAG_TEMPORAL,
                                                                   V.FLAG_TEMPORAL, V.FLAG_DISTRIBUTION]
    ir, info = script.compile_script("s = residue(5:11); v = sdf(s, element('O') and water, 10.0);", topo, lib=emu_lib)
    assert info["v"]["structures"].shape == (7, 10) and info["v"]["structures"][0, 0] == 40


def test_distance_in_context_population(emu_lib, topo):
    """`distance(i, j) in residue(k)` uses indices local to the residue (src/main.cpp:2836-2840); an array of contexts gives a population"""
    ir, info = script.compile_script("d = distance(1, 5) in residue(3); p = distance_min(1:2, element('O')) in resname(\"ALA\")[2:4];",
                                     topo, lib=emu_lib)
    assert info["d"]["a_sets"][0].tolist() == [20] and info["d"]["b_sets"][0].tolist() == [24]
    assert len(info["p"]["a_sets"]) == 3 and info["p"]["a_sets"][1].tolist() == [20, 21]
    assert info["p"]["b_sets"][1].tolist() == [23]                       # the one O of residue 3 (N C C O C H H H C H)
    ev = V.ScriptEval(2, ir)
    assert ev.property_data("d").dim[:2] == (2, 1) and ev.property_data("p").dim[:2] == (2, 3)


def test_script_errors(emu_lib, topo):
    for bad in ("g = rdf(element('X'), all, 5.0);", "v = sdf(all[1:2], all, 5.0);", "d = distance(1, 999999);",
                "d = distance(1, 2) in all;", "d = distance(1, 99) in residue(3);", "g = rdf(all, all 5.0);", "x = frobnicate(3);", "g = rdf(residue(0), all, 5.0);"):
        with pytest.raises((script.ScriptError, V.VmdError)):
            script.compile_script(bad, topo, lib=emu_lib)


def test_script_driven_evaluation_matches_oracle(emu_lib, oracle):
    """config-4/5 style script end to end on the emulator build: blob + waters, sdf + rdf + distances."""
    import cases
    from viamd_amd import _lib as L
    n_blob, n_atoms, box, F = 60, 60 + 900, 30.0, 3
    topo = synth.water_box_topology(n_atoms, n_blob)
    coords = cases.host_frames(oracle, 12, n_atoms, box, F, n_blob)
    ir, info = script.compile_script(
        "s = residue(2:4); v = sdf(s, element('O') and water, 8.0); g = rdf(element('O') and water, not element('H'), 9.0);"
        "d = distance(residue(1), residue(6)); m = distance_max(residue(2), residue(3));", topo, lib=emu_lib)
    ev = V.ScriptEval(F, ir)
    vcell = V.make_unitcell(box)
    assert ev.frame_range(V.MolSystem(n_atoms, mass=topo.mass, unitcell=vcell), V.HostTrajectory(coords, vcell), 0, F)
    ocell = oracle.make_cell(box)
    vol, _ = cases.oracle_sdf(oracle, coords, ocell, info["v"]["structures"], topo.mass, info["v"]["target"], 8.0)
    np.testing.assert_array_equal(ev.property_data("v").counts, vol)
    ref, _ = cases.oracle_rdf(oracle, coords, ocell, info["g"]["ref"], info["g"]["target"], 0.0, 9.0)
    np.testing.assert_array_equal(ev.property_data("g").counts, ref)
    d = cases.oracle_distance(oracle, coords, ocell, topo.mass, info["d"]["a"], info["d"]["b"], L.DIST_COM)
    np.testing.assert_array_equal(ev.property_data("d").values.reshape(F, -1), d)
    m = cases.oracle_distance(oracle, coords, ocell, topo.mass, info["m"]["a"], info["m"]["b"], L.DIST_MAX)
    np.testing.assert_array_equal(ev.property_data("m").values.reshape(F, -1), m)


GOOD_SCRIPTS = [
    "g = rdf(element('O') and water, element('O') and water, 12.0);",
    """# default script shape of src/main.cpp:528
       s1 = resname("ALA")[2:8];
       r = rdf(element('C'), element('H'), 10.0);
       v = sdf(s1, element('H'), 10.0);
       d1 = distance(10, 30);
       d2 = distance_min(residue(3), water and element('O'));
       h = rdf(not element('H'), not element('H'), {2.0, 9.0});""",
    "s = residue(5:11); v = sdf(s, element('O') and water, 10.0)",
    "d = distance(1, 5) in residue(3); p = distance_min(1:2, element('O')) in resname(\"ALA\")[2:4];",
    "q = distance_pair(1:2, 4:6) in residue(10:14); x = distance_max(atom(3), atom(7:9)) in resname('ALA')[1:3];",
    "a = (element('N', 'C') or name('O')) and protein and not residue(1:5); g = rdf(a, water and element('H'), 1.5:7.25); ;",
    "w = water[3:40]; v = sdf(w, all, 4.5); m = distance_min(w[1:2], protein);",
    "t = type('H') and residue(201:300); g = rdf(t, t, 6.0); d = distance(t, label('C'));",
    "v = sdf(residue(1:2) and element('C', 'N'), all, 3.0) ; w = sdf(residue(1) or residue(201), all, 3.0);",
    # BASELINE config 5
    "goo = rdf(element('O') and water, element('O') and water, 12.0);goh = rdf(element('O') and water, element('H') and water, 12.0);"
    "ghv = rdf(not element('H'), not element('H'), 12.0);s = residue(5:11); v = sdf(s, element('O') and water, 10.0);"
    "d1 = distance(1, 1990); d2 = distance(residue(1), residue(200));d3 = distance_min(residue(3), residue(150)); d4 = distance_max(residue(10), residue(20));",
]

BAD_SCRIPTS = ["g = rdf(element('X'), all, 5.0);", "v = sdf(all[1:2], all, 5.0);", "d = distance(1, 999999);", "d = distance(1, 2) in all;",
               "d = distance(1, 99) in residue(3);", "g = rdf(all, all 5.0);", "x = frobnicate(3);", "g = rdf(residue(0), all, 5.0);",
               "g = rdf(all, all, 5.0", "v = sdf(resname('ALA')[1:900], all, 3.0);", "s = element('O) ; g = rdf(s, s, 4.0);", "g = rdf(all, @, 4.0);",
               "w = sdf(resname('ALA', 'HOH')[200:201], all, 3.0);"]


def test_resid_selects_by_the_sequence_number_of_the_file(host_lib, topo):
    """VIAMD emits `in resid(%i)` with md_component_seq_id (the PDB resSeq) next to `in residue(%i)` with the
    residue index + 1 (src/main.cpp:2843-2848).  The two differ whenever resSeq does not start at 1 or restarts per chain;
    without sequence numbers resid() must be rejected, never aliased to residue()."""
    for compile_ in (lambda t, tp: script.compile_script(t, tp, lib=host_lib)[0], lambda t, tp: script.compile_script_native(t, tp, lib=host_lib)):
        with pytest.raises((script.ScriptError, V.VmdError), match="sequence numbers"):
            compile_("d = distance(1, 2) in resid(3);", topo)
    # resSeq starts at 17 in chain A (residues 0..99), restarts at 1 in chain B (residues 100..199), waters count on from 500
    ri = np.asarray(topo.residue_index)
    seq = np.where(ri < 100, ri + 17, np.where(ri < 200, ri - 100 + 1, ri + 300))
    tp = script.Topology(topo.elements, topo.resnames, topo.residue_index, topo.names, mass=topo.mass, residue_seq_id=seq)
    for text, same_as in (("d = distance(1, 2) in resid(20);", "d = distance(1, 2) in residue(4:4) ;"),                  # 20 = 17 + 3 -> residue index 3 (1-based 4) ...
                          ("d = distance_min(1, 2:3) in resid(17:18);", None),
                          ("g = rdf(resid(1:2) and element('C'), water and element('O'), 6.0);", "g = rdf(residue(101:102) and element('C'), water and element('O'), 6.0);"),
                          ("g = rdf(resid(20), all, 5.0);", "g = rdf(residue(4) or residue(120), all, 5.0);")):      # ... and 20 also exists in chain B
        ir_py, info = script.compile_script(text, tp, lib=host_lib)
        ir_c = script.compile_script_native(text, tp, lib=host_lib)
        assert ir_c.fingerprint() == ir_py.fingerprint(), text
        if same_as:
            if "in resid(20)" in text:      # a population of two contexts (chain A residue 4, chain B residue 120)
                assert len(info["d"]["a_sets"]) == 2 and info["d"]["a_sets"][0].tolist() == [30] and info["d"]["a_sets"][1].tolist() == [1190]
            else:
                assert script.compile_script(same_as, tp, lib=host_lib)[0].fingerprint() == ir_py.fingerprint(), text
    with pytest.raises((script.ScriptError, V.VmdError), match="matches no residue"):
        script.compile_script_native("g = rdf(resid(400), all, 5.0);", tp, lib=host_lib)
    with pytest.raises((script.ScriptError, V.VmdError), match="matches no residue"):
        script.compile_script("g = rdf(resid(400), all, 5.0);", tp, lib=host_lib)


def test_native_front_end_matches_the_python_one(host_lib, topo):
    emu_lib = host_lib       # pure host code: the g++ emulator build and the hipcc-built product library
    """vmd_ir_compile_from_source (C++) against viamd_amd/script.py: identical descriptors, i.e. identical IR fingerprints,
    names and flags; the same scripts are rejected."""
    for text in GOOD_SCRIPTS:
        ir_py, _ = script.compile_script(text, topo, lib=emu_lib)
        ir_c = script.compile_script_native(text, topo, lib=emu_lib)
        assert ir_c.property_names() == ir_py.property_names(), text
        assert [ir_c.property_flags(n) for n in ir_c.property_names()] == [ir_py.property_flags(n) for n in ir_py.property_names()]
        assert ir_c.fingerprint() == ir_py.fingerprint(), text
    for text in BAD_SCRIPTS:
        with pytest.raises((script.ScriptError, V.VmdError)):
            script.compile_script_native(text, topo, lib=emu_lib)
        with pytest.raises((script.ScriptError, V.VmdError, ValueError)):
            script.compile_script(text, topo, lib=emu_lib)


# the literal string VIAMD's editor starts with (/root/reference/src/main.cpp:528)
VIAMD_DEFAULT_SCRIPT = ("s1 = resname(\"ALA\")[2:8];\nd1 = distance(10,30);\na1 = angle(2,1,3) in resname(\"ALA\");\n"
                        "r = rdf(element('C'), element('H'), 10.0);\nv = sdf(s1, element('H'), 10.0);\n{lin,plan,iso} = shape_weights(all);")


def test_partial_compile_skips_what_is_outside_the_path(host_lib, topo):
    """one statement outside the subset must not cost the script its rdf / sdf / distance properties.
    VIAMD's default script yields d1, r, v; a1 and {lin,plan,iso} come back as the list the shim's fallback evaluates; C++ and
    Python agree on names, ranges and the fallback text."""
    text = VIAMD_DEFAULT_SCRIPT
    with pytest.raises((script.ScriptError, V.VmdError)):
        script.compile_script_native(text, topo, lib=host_lib)                   # the strict mode still refuses the script as a whole
    ir_py, info, rep_py = script.compile_script(text, topo, lib=host_lib, partial=True)
    ir_c, rep_c = script.compile_script_native(text, topo, lib=host_lib, partial=True)
    assert ir_py.property_names() == ir_c.property_names() == ["d1", "r", "v"]
    assert ir_py.fingerprint() == ir_c.fingerprint()
    strict = script.compile_script("s1 = resname(\"ALA\")[2:8]; d1 = distance(10,30); r = rdf(element('C'), element('H'), 10.0); v = sdf(s1, element('H'), 10.0);",
                                   topo, lib=host_lib)[0]
    assert strict.fingerprint() == ir_c.fingerprint()                            # the same descriptors as the script without the two statements
    assert info["v"]["structures"].shape == (7, 10)
    for rep in (rep_py, rep_c):
        assert [k["names"] for k in rep["skipped"]] == ["a1", "lin,plan,iso"]
        assert [text[k["beg"]:k["end"]] for k in rep["skipped"]] == ["a1 = angle(2,1,3) in resname(\"ALA\")", "{lin,plan,iso} = shape_weights(all)"]
        assert "angle" in rep["skipped"][0]["reason"] and "shape_weights" in rep["skipped"][1]["reason"]
        fb = rep["fallback_source"]
        assert len(fb) == len(text) and fb.count("\n") == text.count("\n")       # offsets unchanged: mdlib's diagnostics still point into the editor's text
        assert "distance" not in fb and "rdf" not in fb and "sdf" not in fb      # no property is evaluated twice
        assert "s1 = resname(\"ALA\")[2:8];" in fb and "a1 = angle(2,1,3) in resname(\"ALA\");" in fb and "{lin,plan,iso} = shape_weights(all);" in fb
    assert rep_py["fallback_source"] == rep_c["fallback_source"]
    assert [(k["names"], k["beg"], k["end"]) for k in rep_py["skipped"]] == [(k["names"], k["beg"], k["end"]) for k in rep_c["skipped"]]


def test_partial_compile_edge_cases(host_lib, topo):
    cases_ = [
        # a statement that uses an identifier of a skipped one goes with it; the next one still compiles
        ("w = within(5.0, protein); g = rdf(w, all, 5.0); h = rdf(water and element('O'), all, 4.0);", ["h"], ["w", "g"]),
        # characters outside the subset, a malformed number, trailing garbage after a complete call, an empty selection
        ("x = 3 * 4 + 2; d = distance(1, 2) ; e = distance(1, 2) + 1; f = rdf(element('X'), all, 5.0); k = distance_max(1:3, 7);", ["d", "k"], ["x", "e", "f"]),
        # a property name that is taken: the second statement is skipped with the library's message
        ("d = distance(1, 2); d = distance(3, 4);", ["d"], ["d"]),
        # nothing understood, nothing lost
        ("{a,b} = shape_weights(all)", [], ["a,b"]),
        ("", [], []),
        # an unbalanced statement swallows the rest of the text: reported as one
        ("g = rdf(all, all, 5.0; h = rdf(all, all, 4.0);", [], ["g"]),
    ]
    # a compiled property a SKIPPED statement uses stays in the fallback's text (mdlib could not compile `x = d * 2` without
    # `d`), transitively; one nobody uses is blanked as before; offsets never move
    text = "d = distance(1, 2); e = distance(2, 3); x = d * 2; g = rdf(all, all, 5.0); {a,b} = f(x, g);"
    for compile_ in (lambda t: script.compile_script(t, topo, lib=host_lib, partial=True)[::2], lambda t: script.compile_script_native(t, topo, lib=host_lib, partial=True)):
        ir_, rep = compile_(text)
        assert ir_.property_names() == ["d", "e", "g"] and [k["names"] for k in rep["skipped"]] == ["x", "a,b"]
        fb = rep["fallback_source"]
        assert len(fb) == len(text)
        assert fb == "d = distance(1, 2);                     x = d * 2; g = rdf(all, all, 5.0); {a,b} = f(x, g);"
    # ... transitively: y is skipped and uses h; h is compiled and (as a selection argument would) mentions nothing else; k uses nothing skipped
    text = "s = residue(1); h = distance(s, 5); k = distance(1, 2); y = h + 1;"
    ir_, rep = script.compile_script_native(text, topo, lib=host_lib, partial=True)
    assert ir_.property_names() == ["h", "k"] and rep["fallback_source"] == "s = residue(1); h = distance(s, 5);                     y = h + 1;"
    for text, compiled, skipped in cases_:
        ir_py, _, rep_py = script.compile_script(text, topo, lib=host_lib, partial=True)
        ir_c, rep_c = script.compile_script_native(text, topo, lib=host_lib, partial=True)
        assert ir_py.property_names() == ir_c.property_names() == compiled, text
        assert [k["names"] for k in rep_py["skipped"]] == [k["names"] for k in rep_c["skipped"]] == skipped, text
        assert [(k["beg"], k["end"]) for k in rep_py["skipped"]] == [(k["beg"], k["end"]) for k in rep_c["skipped"]], text
        assert rep_py["fallback_source"] == rep_c["fallback_source"], text
        assert ir_py.fingerprint() == ir_c.fingerprint(), text
    # every script the strict mode accepts compiles identically in the partial mode, with nothing skipped
    for text in GOOD_SCRIPTS:
        ir_c, rep = script.compile_script_native(text, topo, lib=host_lib, partial=True)
        assert rep["skipped"] == [] and ir_c.fingerprint() == script.compile_script_native(text, topo, lib=host_lib).fingerprint(), text
