"""Mini front-end for the subset of the VIAMD script language the hot path needs.

Stands in for md_script_ir_compile_from_source (/root/reference/src/main.cpp:878) for scripts of the forms VIAMD itself
generates or ships as defaults (src/main.cpp:528, :2817-2858):

    s1 = resname("ALA")[2:8];
    r  = rdf(element('C'), element('H'), 10.0);
    v  = sdf(s1, element('H'), 10.0);
    d1 = distance(10, 30);          # 1-based atom indices (src/main.cpp:2817)
    d2 = distance_min(residue(3), water);

Selections: element('X') | type/name('X') | resname("X") | residue(a:b) | resid(a:b) | atom(a:b) | integer | all | water |
protein, combined with `and`, `or`, `not`, parentheses; `sel[a:b]` slices an array of structures (1-based, inclusive).
The output is a ScriptIR of property descriptors (vmd_ir_add_rdf/_sdf/_distance); nothing is evaluated here.
"""
import re

import numpy as np

from . import _lib as L
from .eval import ScriptIR, VmdError

WATER_RESNAMES = {"HOH", "WAT", "SOL", "TIP3", "TIP4", "SPC", "H2O"}
PROTEIN_RESNAMES = {"ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO",
                    "SER", "THR", "TRP", "TYR", "VAL"}


class ScriptError(ValueError):
    pass


class Topology:
    """What selections are resolved against: per-atom element / name / residue name / residue index (0-based, contiguous)."""

    def __init__(self, elements, resnames=None, residue_index=None, names=None, mass=None, residue_seq_id=None):
        self.elements = np.asarray(elements)
        n = self.elements.size
        # per atom: the residue sequence number the file carries (PDB resSeq, GRO residue number) - what resid() selects by;
        # None = unknown (resid() is then an error, never an alias of residue(); src/main.cpp:2843-2848 emits both forms)
        self.residue_seq_id = None if residue_seq_id is None else np.asarray(residue_seq_id, dtype=np.int64)
        self.resnames = np.asarray(resnames) if resnames is not None else np.array(["UNK"] * n)
        self.residue_index = np.asarray(residue_index, dtype=np.int64) if residue_index is not None else np.zeros(n, np.int64)
        self.names = np.asarray(names) if names is not None else self.elements
        self.mass = mass
        self.num_atoms = n
        self.num_residues = int(self.residue_index.max()) + 1 if n else 0
        # atom range of every residue (residues are contiguous runs of atoms)
        order = np.argsort(self.residue_index, kind="stable")
        self._res_atoms = np.split(order, np.cumsum(np.bincount(self.residue_index, minlength=self.num_residues))[:-1])

    def residue_atoms(self, r):
        return np.sort(self._res_atoms[r])

    def residue_name(self, r):
        return str(self.resnames[self._res_atoms[r][0]])


class Sel:
    """A selection value: a boolean atom mask plus, when it came from a per-residue construct, the list of structures."""

    def __init__(self, mask, structures=None):
        self.mask = mask
        self.structures = structures      # list of index arrays, or None

    def indices(self):
        return np.nonzero(self.mask)[0].astype(np.int32)


_TOKEN = re.compile(r"""\s*(?:(?P<num>\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+|\d+)|(?P<str>'[^']*'|"[^"]*")|(?P<id>[A-Za-z_]\w*)|(?P<op>[=(),;:\[\]{}]))""")


def _tokenize(text, tolerant=False, spans=None):
    """tokens as (kind, text); `spans` (a list) receives the [beg, end) byte range of each.  tolerant: a character outside the subset
    becomes a one-character op token instead of an error (partial compilation)."""
    text = re.sub(r"#[^\n]*", lambda m: " " * len(m.group()), text)       # comments -> blanks: offsets stay those of the source
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            bad = pos + len(text[pos:]) - len(text[pos:].lstrip())
            if not tolerant:
                raise ScriptError(f"unexpected character {text[bad]!r} at offset {pos}")
            out.append(("op", text[bad]))
            if spans is not None:
                spans.append((bad, bad + 1))
            pos = bad + 1
            continue
        pos = m.end()
        kind = m.lastgroup
        out.append((kind, m.group(kind)))
        if spans is not None:
            spans.append((m.start(kind), m.end(kind)))
    return out


class _Parser:
    def __init__(self, tokens, topo, env, ctx=None):
        # ctx: atom indices of the evaluation context (`... in residue(3)`): integer / atom() indices are then relative to it
        # (src/main.cpp:2836-2840: `distance(i, j) in residue(k)` uses indices local to the residue) and every mask is clipped to it
        self.t, self.i, self.topo, self.env, self.ctx = tokens, 0, topo, env, ctx
        self.ctx_mask = None
        if ctx is not None:
            self.ctx_mask = np.zeros(topo.num_atoms, bool)
            self.ctx_mask[ctx] = True

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def take(self, value=None, kind=None):
        k, v = self.peek()
        if k is None or (value is not None and v != value) or (kind is not None and k != kind):
            raise ScriptError(f"expected {value or kind}, found {v!r}")
        self.i += 1
        return v

    def accept(self, value):
        if self.peek()[1] == value:
            self.i += 1
            return True
        return False

    # range a[:b], 1-based inclusive -> python slice bounds (0-based, exclusive)
    def range_(self):
        a = int(self.take(kind="num"))
        b = a
        if self.accept(":"):
            b = int(self.take(kind="num"))
        if a < 1 or b < a:
            raise ScriptError(f"bad range {a}:{b} (script indices are 1-based)")
        return a - 1, b

    def sel_or(self):
        s = self.sel_and()
        while self.accept("or"):
            s = Sel(s.mask | self.sel_and().mask)
        return s

    def sel_and(self):
        s = self.sel_not()
        while self.accept("and"):
            r = self.sel_not()
            structs = None
            if s.structures is not None:      # per-structure intersection keeps the array shape
                structs = [st[r.mask[st]] for st in s.structures]
            s = Sel(s.mask & r.mask, structs)
        return s

    def sel_not(self):
        if self.accept("not"):
            return Sel(~self.sel_not().mask)
        return self.sel_postfix()

    def sel_postfix(self):
        s = self.sel_atom()
        while self.accept("["):
            a, b = self.range_()
            self.take("]")
            if s.structures is None:
                raise ScriptError("[a:b] applies to an array of structures (resname(...), residue(...))")
            if b > len(s.structures):
                raise ScriptError(f"slice [{a + 1}:{b}] exceeds the {len(s.structures)} structures of the selection")
            st = s.structures[a:b]
            mask = np.zeros(self.topo.num_atoms, bool)
            for x in st:
                mask[x] = True
            s = Sel(mask, st)
        return s

    def _residues(self, pred):
        topo = self.topo
        st = [topo.residue_atoms(r) for r in range(topo.num_residues) if pred(r)]
        mask = np.zeros(topo.num_atoms, bool)
        for x in st:
            mask[x] = True
        return Sel(mask, st)

    def _atom_range(self, a, b):
        topo = self.topo
        mask = np.zeros(topo.num_atoms, bool)
        if self.ctx is not None:
            if b > len(self.ctx):
                raise ScriptError(f"atom index {b} out of range (the context has {len(self.ctx)} atoms)")
            mask[self.ctx[a:b]] = True
        else:
            if b > topo.num_atoms:
                raise ScriptError(f"atom index {b} out of range (system has {topo.num_atoms} atoms)")
            mask[a:b] = True
        return mask

    def sel_atom(self):
        s = self._sel_atom()
        if self.ctx_mask is not None and self.peek()[1] != "[":
            structs = None if s.structures is None else [st[self.ctx_mask[st]] for st in s.structures]
            s = Sel(s.mask & self.ctx_mask, structs)
        return s

    def _sel_atom(self):
        k, v = self.peek()
        topo = self.topo
        if v == "(":
            self.take("(")
            s = self.sel_or()
            self.take(")")
            return s
        if k == "num":
            a, b = self.range_()
            return Sel(self._atom_range(a, b))
        if k != "id":
            raise ScriptError(f"unexpected token {v!r} in selection")
        self.i += 1
        if v in self.env and isinstance(self.env[v], Sel):
            return self.env[v]
        if v == "all":
            return Sel(np.ones(topo.num_atoms, bool))
        if v == "water":
            return self._residues(lambda r: topo.residue_name(r).upper() in WATER_RESNAMES)
        if v == "protein":
            return self._residues(lambda r: topo.residue_name(r).upper() in PROTEIN_RESNAMES)
        if v in ("element", "type", "name", "label", "resname"):
            self.take("(")
            names = [self.take(kind="str")[1:-1]]
            while self.accept(","):
                names.append(self.take(kind="str")[1:-1])
            self.take(")")
            if v == "resname":
                return self._residues(lambda r: topo.residue_name(r) in names)
            arr = topo.elements if v == "element" else topo.names
            return Sel(np.isin(arr, names))
        if v == "resid":
            self.take("(")
            a = int(self.take(kind="num"))
            b = a
            if self.accept(":"):
                b = int(self.take(kind="num"))
            self.take(")")
            if b < a:
                raise ScriptError(f"bad range {a}:{b}")
            if topo.residue_seq_id is None:
                raise ScriptError("resid(): the topology carries no residue sequence numbers (vmd_topology_t.residue_seq_id); "
                                  "use residue() for the 1-based residue index")
            seq = [int(topo.residue_seq_id[topo._res_atoms[r][0]]) for r in range(topo.num_residues)]
            s = self._residues(lambda r: a <= seq[r] <= b)
            if not s.structures:
                raise ScriptError(f"resid({a}:{b}) matches no residue")
            return s
        if v in ("residue", "atom"):
            self.take("(")
            a, b = self.range_()
            self.take(")")
            if v == "atom":
                return Sel(self._atom_range(a, b))
            if b > topo.num_residues:
                raise ScriptError(f"{v}({b}) out of range (system has {topo.num_residues} residues)")
            return self._residues(lambda r: a <= r < b)
        if self.peek()[1] == "(":
            raise ScriptError(f"unsupported function {v!r} (outside the rdf / sdf / distance path)")
        raise ScriptError(f"unknown identifier {v!r}")

    def number(self):
        return float(self.take(kind="num"))


_FUNCS = {"rdf", "sdf", "distance", "distance_min", "distance_max", "distance_pair"}
_DIST_KIND = {"distance": L.DIST_COM, "distance_min": L.DIST_MIN, "distance_max": L.DIST_MAX, "distance_pair": L.DIST_PAIR}


def compile_script(text, topo, lib=None, partial=False):
    """Returns (ScriptIR, info) where info[name] = dict(kind=..., plus the resolved index arrays).

    partial=True (vmd_ir_compile_from_source_partial): statements outside the subset are reported instead of failing the script;
    returns (ScriptIR, info, report) with report = dict(skipped=[dict(names, beg, end, reason)], fallback_source=text with the compiled
    property statements blanked out).  VIAMD's default script (src/main.cpp:528) then yields d1, r, v and reports a1 and lin,plan,iso."""
    ir = ScriptIR(lib)
    env, info = {}, {}
    spans = []
    toks = _tokenize(text, tolerant=partial, spans=spans)
    p = _Parser(toks, topo, env)
    skipped = []
    fallback = list(text)
    blanked, skipped_toks = [], []      # compiled property statements [name, tok_first, tok_end, beg, end, kept]; token ranges of the skipped ones
    while p.peek()[0] is not None:
        if p.accept(";"):
            continue
        first = p.i
        last, depth = first, 0
        while last < len(toks):
            v = toks[last][1] if toks[last][0] == "op" else None
            if v in ("(", "[", "{"):
                depth += 1
            elif v in (")", "]", "}"):
                depth -= 1
            elif v == ";" and depth <= 0:
                break
            last += 1
        names = ""
        try:
            if p.peek()[1] == "{" and p.peek()[0] == "op":
                p.take("{")
                names = p.take(kind="id")
                while p.accept(","):
                    names += "," + p.take(kind="id")
                p.take("}")
                p.take("=")
                k, v = p.peek()
                raise ScriptError(f"unsupported {'function' if k == 'id' else 'expression'} {v!r} (outside the rdf / sdf / distance path)")
            name = p.take(kind="id")
            names = name
            p.take("=")
            commit, is_property = _statement(p, name, topo, env, ir, info)
            if p.peek()[0] is not None and p.peek()[1] != ";":
                raise ScriptError(f"expected ;, found {p.peek()[1]!r}")
            commit()
            if p.peek()[0] is not None:
                p.take(";")
            if is_property and partial:
                e = spans[last][1] if last < len(toks) else spans[last - 1][1]
                for c in range(spans[first][0], min(e, len(fallback))):
                    if fallback[c] != "\n":
                        fallback[c] = " "
                blanked.append([names, first, last, spans[first][0], min(e, len(fallback)), False])
        except (ValueError, VmdError) as e:   # ScriptError, int() / float() on a malformed number, a descriptor the library refuses
            if not partial:
                raise
            send = spans[last - 1][1] if last > first else spans[first][1]
            skipped.append(dict(names=names or toks[first][1], beg=spans[first][0], end=send, reason=str(e)))
            skipped_toks.append((first, last))
            p.i = last + 1 if last < len(toks) else last
    if partial:
        # a compiled property that a skipped statement uses stays in the fallback's text (the twin of vmd_script.cpp)
        used = set()

        def collect(a, b):
            k = a
            for q in range(a, min(b, len(toks))):
                if toks[q][0] == "op" and toks[q][1] == "=":
                    k = q + 1
                    break
            used.update(toks[q][1] for q in range(k, min(b, len(toks))) if toks[q][0] == "id")

        for a, b in skipped_toks:
            collect(a, b)
        changed = True
        while changed:
            changed = False
            for item in blanked:
                if item[5] or item[0] not in used:
                    continue
                item[5] = changed = True
                fallback[item[3]:item[4]] = text[item[3]:item[4]]
                collect(item[1], item[2])
        return ir, info, dict(skipped=skipped, fallback_source="".join(fallback))
    return ir, info


def _statement(p, name, topo, env, ir, info):
    """parses the right-hand side of `name = ...` up to (not including) the ';'.  Returns (commit, is_property): nothing is added to the
    IR or to the identifiers before commit() runs, so a statement that fails half way leaves nothing behind."""
    k, v = p.peek()
    if not (k == "id" and v in _FUNCS):
        sel = p.sel_or()
        return (lambda: env.__setitem__(name, sel)), False
    p.i += 1
    p.take("(")
    if v == "rdf":
        ref = p.sel_or(); p.take(",")
        tgt = p.sel_or(); p.take(",")
        if p.accept("{"):
            rmin = p.number(); p.take(","); rmax = p.number(); p.take("}")
        else:
            rmin, rmax = 0.0, p.number()
            if p.accept(":"):
                rmin, rmax = rmax, p.number()
        p.take(")")
        a, b = ref.indices(), tgt.indices()
        if a.size == 0 or b.size == 0:
            raise ScriptError(f"{name}: empty selection")

        def commit():
            ir.add_rdf(name, a, b, (rmin, rmax))
            info[name] = dict(kind="rdf", ref=a, target=b, rmin=rmin, rmax=rmax)
        return commit, True
    if v == "sdf":
        ref = p.sel_or(); p.take(",")
        tgt = p.sel_or(); p.take(",")
        cutoff = p.number()
        p.take(")")
        structs = ref.structures if ref.structures is not None else [ref.indices()]
        sizes = {len(s) for s in structs}
        if len(sizes) != 1 or 0 in sizes:
            raise ScriptError(f"{name}: sdf reference structures must be non-empty and of equal size, got sizes {sorted(sizes)}")
        st = np.stack([np.asarray(s, np.int32) for s in structs])

        def commit():
            ir.add_sdf(name, st, tgt.indices(), cutoff)
            info[name] = dict(kind="sdf", structures=st, target=tgt.indices(), cutoff=cutoff)
        return commit, True
    # the arguments may be followed by `in <contexts>`: find the closing parenthesis first
    start, depth, j = p.i, 1, p.i
    while depth:
        if j >= len(p.t):
            raise ScriptError(f"{name}: missing ')'")
        depth += {"(": 1, ")": -1}.get(p.t[j][1], 0) if p.t[j][0] == "op" else 0
        j += 1
    if j < len(p.t) and p.t[j] == ("id", "in"):
        q = _Parser(p.t, topo, env)
        q.i = j + 1
        ctx = q.sel_or()
        if ctx.structures is None or not ctx.structures:
            raise ScriptError(f"{name}: `in` needs an array of structures (residue(...), resname(...))")
        a_sets, b_sets = [], []
        for st in ctx.structures:
            r = _Parser(p.t, topo, env, ctx=np.asarray(st))
            r.i = start
            a = r.sel_or(); r.take(","); b = r.sel_or(); r.take(")")
            if a.indices().size == 0 or b.indices().size == 0:
                raise ScriptError(f"{name}: empty selection inside a context")
            a_sets.append(a.indices()); b_sets.append(b.indices())
        p.i = q.i

        def commit():
            ir.add_distance_population(name, a_sets, b_sets, _DIST_KIND[v])
            info[name] = dict(kind=v, a_sets=a_sets, b_sets=b_sets)
        return commit, True
    a = p.sel_or(); p.take(",")
    b = p.sel_or()
    p.take(")")

    def commit():
        ir.add_distance(name, a.indices(), b.indices(), _DIST_KIND[v])
        info[name] = dict(kind=v, a=a.indices(), b=b.indices())
    return commit, True


def compile_script_native(text, topo, lib=None, partial=False):
    """The same front-end in C++ (vmd_ir_compile_from_source, viamd_amd/csrc/vmd_script.cpp): what a C / C++ host calls.
    Returns a ScriptIR; raises ScriptError with the library's message."""
    import ctypes as C
    ir = ScriptIR(lib)
    n = topo.num_atoms

    def strings(arr):
        return (C.c_char_p * n)(*[str(v).encode() for v in arr])

    el, nm, rn = strings(topo.elements), strings(topo.names), strings(topo.resnames)
    ri = np.ascontiguousarray(topo.residue_index, np.int32)
    sq = None if topo.residue_seq_id is None else np.ascontiguousarray(topo.residue_seq_id, np.int32)
    tc = L.TopologyC(n, el, nm, rn, ri.ctypes.data_as(L.c_int32_p), sq.ctypes.data_as(L.c_int32_p) if sq is not None else None)
    if not partial:
        if not ir.lib.vmd_ir_compile_from_source(ir.h, text.encode(), C.byref(tc)):
            raise ScriptError(ir.lib.last_error())
        return ir
    rep = C.c_void_p()
    if not ir.lib.vmd_ir_compile_from_source_partial(ir.h, text.encode(), C.byref(tc), C.byref(rep)):
        raise ScriptError(ir.lib.last_error())
    try:
        n = ir.lib.vmd_script_report_skipped_count(rep)
        items = ir.lib.vmd_script_report_skipped(rep)
        skipped = [dict(names=items[k].names.decode(), beg=items[k].beg, end=items[k].end, reason=items[k].reason.decode()) for k in range(n)]
        report = dict(skipped=skipped, fallback_source=ir.lib.vmd_script_report_fallback_source(rep).decode())
    finally:
        ir.lib.vmd_script_report_free(rep)
    return ir, report
