"""Random scripts through both front-ends (viamd_amd/script.py and the C++ vmd_ir_compile_from_source): both must accept and
produce the same IR fingerprint, or both must reject.  Every script also goes through the PARTIAL mode of both (round 5), salted with statements
outside the subset (angle, shape_weights, arithmetic, stray characters): same compiled properties, same skipped names and source ranges, same
fallback text; whatever the strict mode accepts the partial mode compiles identically with nothing skipped.
usage: python scripts/fuzz_script.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest
import viamd_amd as V
from viamd_amd import script, synth

lib = V.VmdLib(conftest.build_emu())
topo = synth.water_box_topology(300 + 300, n_blob=300)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rint(lo, hi):
    return int(rng.integers(lo, hi + 1))


def rng_range(hi):
    a = rint(0, hi + 2)
    return str(a) if rng.random() < 0.4 else f"{a}:{rint(max(0, a - 1), hi + 3)}"


def atom_sel(depth=0):
    r = rng.random()
    if depth > 2 or r < 0.55:
        return rng.choice(["all", "water", "protein", "element('O')", "element('H', 'C')", "name('N')", 'resname("ALA")', "resname('HOH')",
                           f"residue({rng_range(130)})", f"atom({rng_range(600)})", rng_range(600),
                           f'resname("ALA")[{rng_range(30)}]', f"water[{rng_range(100)}]", "type('Q')", "s0"])
    if r < 0.7:
        return f"not {atom_sel(depth + 1)}"
    if r < 0.85:
        return f"({atom_sel(depth + 1)} {rng.choice(['and', 'or'])} {atom_sel(depth + 1)})"
    return f"{atom_sel(depth + 1)} {rng.choice(['and', 'or'])} {atom_sel(depth + 1)}"


def statement(i):
    r = rng.random()
    if r < 0.15:
        return f"s{i} = {atom_sel()}"
    if r < 0.45:
        cut = rng.choice(["7.5", "{1.0, 6.25}", "0.5:9", "12", ".5"])
        return f"g{i} = rdf({atom_sel()}, {atom_sel()}, {cut})"
    if r < 0.6:
        return f"v{i} = sdf({atom_sel()}, {atom_sel()}, {rng.choice(['4.0', '10', '3.25e0'])})"
    fn = rng.choice(["distance", "distance_min", "distance_max", "distance_pair"])
    ctx = "" if rng.random() < 0.5 else f" in {rng.choice(['residue(' + rng_range(130) + ')', 'resname(' + chr(34) + 'ALA' + chr(34) + ')[' + rng_range(30) + ']', 'water[' + rng_range(50) + ']', 'all'])}"
    return f"d{i} = {fn}({atom_sel()}, {atom_sel()}){ctx}"


def foreign(i):
    return rng.choice([f"a{i} = angle(2,1,3) in resname(\"ALA\")", "{lin,plan,iso} = shape_weights(all)", f"x{i} = 3 * 4 + 2", f"w{i} = within(5.0, protein)",
                       f"q{i} = rdf(w{max(i - 1, 0)}, all, 5.0)", f"z{i} = rmsd(protein) @ 2", f"y{i} = distance(1, 2) + 1", f"{{a,b}} = plane(resname('ALA'))", f"u{i} = 'unterminated"])


agree_ok = agree_err = partial_ok = 0
for it in range(ncases):
    text = "s0 = residue(2:6); " + "; ".join(statement(i) for i in range(rint(1, 4))) + rng.choice([";", "", " ;  # tail comment"])
    res = []
    for fn in (lambda: script.compile_script(text, topo, lib=lib)[0], lambda: script.compile_script_native(text, topo, lib=lib)):
        try:
            ir = fn()
            res.append(("ok", ir.fingerprint(), tuple(ir.property_names())))
        except (script.ScriptError, V.VmdError, ValueError) as e:
            res.append(("err", str(e)[:80]))
    if res[0][0] != res[1][0] or (res[0][0] == "ok" and res[0] != res[1]):
        print("DISAGREE:", text, res)
    elif res[0][0] == "ok":
        agree_ok += 1
    else:
        agree_err += 1
    # partial mode: the same statements with a few foreign ones mixed in
    parts = ["s0 = residue(2:6)"] + [statement(i) if rng.random() < 0.7 else foreign(i) for i in range(rint(1, 5))]
    ptext = "; ".join(parts) + rng.choice([";", ""])
    try:
        ir_py, _, rep_py = script.compile_script(ptext, topo, lib=lib, partial=True)
        ir_c, rep_c = script.compile_script_native(ptext, topo, lib=lib, partial=True)
        same = (ir_py.fingerprint() == ir_c.fingerprint() and ir_py.property_names() == ir_c.property_names() and
                [(k["names"], k["beg"], k["end"]) for k in rep_py["skipped"]] == [(k["names"], k["beg"], k["end"]) for k in rep_c["skipped"]] and
                rep_py["fallback_source"] == rep_c["fallback_source"] and len(rep_c["fallback_source"]) == len(ptext))
        if res[0][0] == "ok":      # the strict text above: partial mode must agree with strict mode, nothing skipped
            ir_s, rep_s = script.compile_script_native(text, topo, lib=lib, partial=True)
            same = same and rep_s["skipped"] == [] and ir_s.fingerprint() == res[1][1]
        if not same:
            print("PARTIAL DISAGREE:", ptext, rep_py["skipped"], rep_c["skipped"])
        else:
            partial_ok += 1
    except Exception as e:          # noqa: BLE001
        print("PARTIAL ERROR:", ptext, repr(e)[:200])
print(f"{ncases} scripts: {agree_ok} accepted identically, {agree_err} rejected by both; partial mode: {partial_ok} of {ncases} agree (properties, skipped names and ranges, fallback text)")
