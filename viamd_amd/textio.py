"""Text trajectory decoders for the staging step in front of the hot path: multi-frame XYZ / XMOL and LAMMPS
dump files.  VIAMD reads both through mdlib (`md_xyz_system_init_from_file`, `md_lammps_trajectory_attach_from_file`,
/root/reference/src/loader.cpp:129-136, 147-148).  They return plain arrays; `viamd_amd.HostTrajectory(coords, cells)` or
`PinnedHostTrajectory.upload` stages them for the evaluator."""
import numpy as np

from .eval import make_unitcell


def read_xyz(path):
    """Multi-frame XYZ: [natoms, comment, natoms x (element x y z ...)] repeated.  An extended-XYZ `Lattice="ax ay az bx by bz
    cx cy cz"` in the comment line becomes the unit cell (lower-triangular form required).
    Returns (coords float32 [F, 3, N], elements [N], cells list[F])."""
    frames, cells, elements = [], [], None
    with open(path) as f:
        while True:
            head = f.readline()
            if not head.strip():
                if head == "":
                    break
                continue
            n = int(head.split()[0])
            comment = f.readline()
            xyz = np.empty((3, n), np.float32)
            elems = []
            for i in range(n):
                t = f.readline().split()
                if len(t) < 4:
                    raise ValueError(f"{path}: frame {len(frames)}: atom line {i} is incomplete")
                elems.append(t[0])
                xyz[:, i] = (float(t[1]), float(t[2]), float(t[3]))
            if elements is None:
                elements = elems
            elif len(elems) != len(elements):
                raise ValueError(f"{path}: frame {len(frames)} has {len(elems)} atoms, the first frame {len(elements)}")
            frames.append(xyz)
            cells.append(_lattice_cell(comment))
    if not frames:
        raise ValueError(f"{path}: no frames")
    return np.stack(frames), np.array(elements), cells


def _lattice_cell(comment):
    key = 'Lattice="'
    i = comment.find(key)
    if i < 0:
        return make_unitcell(None)
    v = [float(t) for t in comment[i + len(key):comment.index('"', i + len(key))].split()]
    if len(v) != 9 or abs(v[1]) > 1e-6 or abs(v[2]) > 1e-6 or abs(v[5]) > 1e-6:
        raise ValueError("extended XYZ lattice must be lower triangular: a=(x,0,0), b=(xy,y,0), c=(xz,yz,z)")
    return make_unitcell((v[0], v[4], v[8]), tilt=(v[3], v[6], v[7]))


def write_xyz(path, coords, elements, cells=None):
    coords = np.asarray(coords)
    with open(path, "w") as f:
        for m in range(coords.shape[0]):
            f.write(f"{coords.shape[2]}\n")
            c = cells[m] if cells is not None else None
            if c is not None and c.flags:
                f.write(f'Lattice="{c.x!r} 0.0 0.0 {c.xy!r} {c.y!r} 0.0 {c.xz!r} {c.yz!r} {c.z!r}" frame={m}\n')
            else:
                f.write(f"frame {m}\n")
            for i in range(coords.shape[2]):
                f.write(f"{elements[i]} {coords[m, 0, i]:.6f} {coords[m, 1, i]:.6f} {coords[m, 2, i]:.6f}\n")


def read_lammps_dump(path):
    """LAMMPS text dump (`dump custom`): per frame ITEM: TIMESTEP / NUMBER OF ATOMS / BOX BOUNDS [xy xz yz] / ATOMS <columns>.
    Columns: id (frames are sorted by it), type, and one of x y z | xu yu zu | xs ys zs (scaled by the cell).
    Returns (coords float32 [F, 3, N], types int [N], cells list[F], timesteps list[F])."""
    frames, cells, steps, types = [], [], [], None
    with open(path) as f:
        line = f.readline()
        while line:
            if not line.startswith("ITEM: TIMESTEP"):
                line = f.readline()
                continue
            steps.append(int(f.readline()))
            assert f.readline().startswith("ITEM: NUMBER OF ATOMS")
            n = int(f.readline())
            bounds = f.readline()
            assert bounds.startswith("ITEM: BOX BOUNDS")
            tri = "xy" in bounds.split()
            periodic = [t for t in bounds.split() if len(t) == 2 and t[0] in "pfsm" and t[1] in "pfsm"]
            rows = [[float(v) for v in f.readline().split()] for _ in range(3)]
            xy, xz, yz = (rows[0][2], rows[1][2], rows[2][2]) if tri else (0.0, 0.0, 0.0)
            # bounds of a triclinic box are those of its bounding box: undo (LAMMPS manual, "triclinic")
            xlo = rows[0][0] - min(0.0, xy, xz, xy + xz); xhi = rows[0][1] - max(0.0, xy, xz, xy + xz)
            ylo = rows[1][0] - min(0.0, yz); yhi = rows[1][1] - max(0.0, yz)
            zlo, zhi = rows[2][0], rows[2][1]
            flags = sum((1 << a) for a in range(3) if a < len(periodic) and periodic[a] == "pp") if periodic else 7
            cell = make_unitcell((xhi - xlo, yhi - ylo, zhi - zlo), flags, (xy, xz, yz))
            head = f.readline()
            assert head.startswith("ITEM: ATOMS")
            cols = head.split()[2:]
            data = np.array([f.readline().split() for _ in range(n)], dtype=np.float64)
            if "id" in cols:
                data = data[np.argsort(data[:, cols.index("id")], kind="stable")]
            for names, scaled in ((("x", "y", "z"), False), (("xu", "yu", "zu"), False), (("xs", "ys", "zs"), True)):
                if all(c in cols for c in names):
                    xyz = np.stack([data[:, cols.index(c)] for c in names])
                    if scaled:
                        s = xyz
                        xyz = np.stack([xlo + s[0] * (xhi - xlo) + s[1] * xy + s[2] * xz, ylo + s[1] * (yhi - ylo) + s[2] * yz,
                                        zlo + s[2] * (zhi - zlo)])
                    break
            else:
                raise ValueError(f"{path}: no coordinate columns among {cols}")
            if types is None:
                types = data[:, cols.index("type")].astype(np.int64) if "type" in cols else np.zeros(n, np.int64)
            elif xyz.shape[1] != types.size:
                raise ValueError(f"{path}: the atom count changes between frames")
            frames.append(xyz.astype(np.float32))
            cells.append(cell)
            line = f.readline()
    if not frames:
        raise ValueError(f"{path}: no frames")
    return np.stack(frames), types, cells, steps


def read_gro(path):
    """GROMACS .gro, one or many frames (VIAMD: md_gro_system_init_from_file, src/loader.cpp:125-126).  Fixed columns
    `%5d%-5s%5s%5d%8.3f%8.3f%8.3f`, nanometres (converted to Angstrom here); box line `v1x v2y v3z [v1y v1z v2x v2z v3x v3y]`.
    Returns (coords float32 [F, 3, N] in Angstrom, dict(resid, resname, name) of the first frame, cells list[F])."""
    frames, cells, meta = [], [], None
    with open(path) as f:
        while True:
            title = f.readline()
            if title == "":
                break
            count = f.readline()
            if not count.strip():
                break
            n = int(count)
            xyz = np.empty((3, n), np.float32)
            resid, resname, name = [], [], []
            for i in range(n):
                line = f.readline()
                if len(line) < 44:
                    raise ValueError(f"{path}: frame {len(frames)}: atom line {i} is too short")
                if meta is None:
                    resid.append(int(line[0:5])); resname.append(line[5:10].strip()); name.append(line[10:15].strip())
                # the coordinate columns may be wider than 8.3 (higher precision files): split the rest evenly in 3 or 6 fields
                rest = line[20:].rstrip("\n")
                ncol = 6 if len(rest.split()) >= 6 else 3
                w = len(rest) // ncol if len(rest) % ncol == 0 else 8
                xyz[:, i] = [10.0 * float(rest[k * w:(k + 1) * w]) for k in range(3)]
            box = [10.0 * float(t) for t in f.readline().split()]
            if len(box) == 3:
                cell = make_unitcell(box)
            elif len(box) == 9:
                if abs(box[3]) > 1e-6 or abs(box[4]) > 1e-6 or abs(box[6]) > 1e-6:
                    raise ValueError(f"{path}: box vectors must be lower triangular (v1 along x, v2 in the xy plane)")
                cell = make_unitcell(box[:3], tilt=(box[5], box[7], box[8]))
            else:
                raise ValueError(f"{path}: bad box line")
            if meta is None:
                meta = dict(resid=np.array(resid), resname=np.array(resname), name=np.array(name))
            elif xyz.shape[1] != meta["resid"].size:
                raise ValueError(f"{path}: the atom count changes between frames")
            frames.append(xyz)
            cells.append(cell)
    if not frames:
        raise ValueError(f"{path}: no frames")
    return np.stack(frames), meta, cells


def write_gro(path, coords, resid, resname, name, cells):
    """coords in Angstrom [F, 3, N]; one Unitcell per frame."""
    coords = np.asarray(coords)
    with open(path, "w") as f:
        for m in range(coords.shape[0]):
            f.write(f"frame t= {float(m):.5f}\n{coords.shape[2]:5d}\n")
            for i in range(coords.shape[2]):
                f.write("%5d%-5s%5s%5d%8.3f%8.3f%8.3f\n" % (resid[i] % 100000, resname[i][:5], name[i][:5], (i + 1) % 100000,
                                                            coords[m, 0, i] / 10.0, coords[m, 1, i] / 10.0, coords[m, 2, i] / 10.0))
            c = cells[m]
            if c.xy == 0.0 and c.xz == 0.0 and c.yz == 0.0:
                f.write("%10.5f%10.5f%10.5f\n" % (c.x / 10, c.y / 10, c.z / 10))
            else:
                f.write("%10.5f%10.5f%10.5f%10.5f%10.5f%10.5f%10.5f%10.5f%10.5f\n" % (c.x / 10, c.y / 10, c.z / 10, 0, 0, c.xy / 10, 0, c.xz / 10, c.yz / 10))


def _cif_tokens(line):
    """Whitespace-separated mmCIF tokens with '...' / "..." quoting."""
    out, i, n = [], 0, len(line)
    while i < n:
        c = line[i]
        if c.isspace():
            i += 1
        elif c in "'\"" and (i == 0 or line[i - 1].isspace()):
            j = i + 1
            while j < n and not (line[j] == c and (j + 1 == n or line[j + 1].isspace())):
                j += 1
            out.append(line[i + 1:j])
            i = j + 1
        elif c == "#":
            break
        else:
            j = i
            while j < n and not line[j].isspace():
                j += 1
            out.append(line[i:j])
            i = j
    return out


def read_mmcif(path):
    """PDBx/mmCIF (VIAMD: md_mmcif_system_init_from_file, src/loader.cpp:137-138): the `_atom_site` loop (Cartn_x/y/z,
    type_symbol, label_atom_id, label_comp_id, label_asym_id / auth_asym_id, label_seq_id / auth_seq_id, pdbx_PDB_model_num) and
    the `_cell.*` items.  Returns (coords float32 [F, 3, N] - one frame per model -, dict(element, name, resname, chain, resid),
    cell_params (a, b, c, alpha, beta, gamma) or None)."""
    cell, cols, rows = {}, [], []
    loop_cols, in_header = None, False          # loop_cols: column names of the loop being read (None outside a loop)
    with open(path) as f:
        for raw in f:
            line = raw.rstrip("\n")
            s = line.strip()
            if not s or s.startswith("#"):
                loop_cols = None                                            # blank line / comment ends a loop
                continue
            if s == "loop_":
                loop_cols, in_header = [], True
                continue
            if s.startswith("data_"):
                loop_cols = None
                continue
            if s.startswith("_"):
                if loop_cols is not None and in_header:
                    loop_cols.append(s.split()[0])
                    continue
                loop_cols = None                                            # a key-value item: any loop before it is over
                if s.startswith("_cell."):
                    t = _cif_tokens(s)
                    if len(t) >= 2:
                        cell[t[0][6:]] = t[1]
                continue
            if loop_cols is not None:                                       # a data row of the current loop
                in_header = False
                if loop_cols and loop_cols[0].startswith("_atom_site."):
                    if not cols:
                        cols = [c[11:] for c in loop_cols]
                    t = _cif_tokens(line)
                    if len(t) == len(cols):
                        rows.append(t)
    if not rows:
        raise ValueError(f"{path}: no _atom_site records")
    ci = {c: k for k, c in enumerate(cols)}
    for need in ("Cartn_x", "Cartn_y", "Cartn_z"):
        if need not in ci:
            raise ValueError(f"{path}: _atom_site.{need} is missing")

    def col(*names, default="."):
        for nm in names:
            if nm in ci:
                return [r[ci[nm]] for r in rows]
        return [default] * len(rows)

    model = col("pdbx_PDB_model_num", default="1")
    models = sorted(set(model), key=lambda m: int(m))
    xyz = np.array([[float(r[ci["Cartn_x"]]), float(r[ci["Cartn_y"]]), float(r[ci["Cartn_z"]])] for r in rows], np.float32)
    model = np.array(model)
    frames = [xyz[model == m].T for m in models]
    n = frames[0].shape[1]
    if any(fr.shape[1] != n for fr in frames):
        raise ValueError(f"{path}: the models have different atom counts")
    first = model == models[0]
    pick = lambda v: np.array(v)[first]
    seq = pick(col("label_seq_id", "auth_seq_id", default="0"))
    if "auth_seq_id" in ci:
        auth = pick(col("auth_seq_id"))
        seq = np.where(np.isin(seq, (".", "?")), auth, seq)           # waters / ligands carry no label_seq_id
    elem = pick(col("type_symbol", "label_atom_id"))
    meta = dict(element=np.array([e.capitalize() for e in elem]), name=pick(col("label_atom_id", "auth_atom_id")),
                resname=pick(col("label_comp_id", "auth_comp_id")), chain=pick(col("label_asym_id", "auth_asym_id")),
                resid=np.array([int(v) if v.lstrip("-").isdigit() else 0 for v in seq]))
    params = None
    if all(k in cell for k in ("length_a", "length_b", "length_c")):
        g = lambda k, d: float(cell.get(k, d).split("(")[0])
        params = (g("length_a", "0"), g("length_b", "0"), g("length_c", "0"), g("angle_alpha", "90"), g("angle_beta", "90"), g("angle_gamma", "90"))
    return np.stack(frames), meta, params


def cell_from_parameters(a, b, c, alpha=90.0, beta=90.0, gamma=90.0):
    """(a, b, c, alpha, beta, gamma) -> Unitcell with a = (x,0,0), b = (xy,y,0), c = (xz,yz,z) (the CRYST1 / _cell convention)."""
    if not (a > 0 and b > 0 and c > 0):
        return make_unitcell(None)
    ca, cb, cg = (0.0 if ang == 90.0 else float(np.cos(np.deg2rad(ang))) for ang in (alpha, beta, gamma))
    xy, xz = b * cg, c * cb
    ly = float(np.sqrt(b * b - xy * xy))
    yz = (b * c * ca - xy * xz) / ly
    lz = float(np.sqrt(c * c - xz * xz - yz * yz))
    return make_unitcell((a, ly, lz), tilt=tuple(0.0 if abs(v) < 1e-6 else float(v) for v in (xy, xz, yz)))


LAMMPS_ATOM_STYLES = {          # columns after the atom id: where type, (molecule), x are (md_lammps_atom_format_*, loader.cpp:80-88)
    "atomic": dict(type=1, mol=None, x=2), "charge": dict(type=1, mol=None, x=3), "molecular": dict(type=2, mol=1, x=3),
    "full": dict(type=2, mol=1, x=4), "bond": dict(type=2, mol=1, x=3), "angle": dict(type=2, mol=1, x=3),
}


def read_lammps_data(path, atom_style=None):
    """LAMMPS data file (VIAMD: md_lammps_system_init_from_file, src/loader.cpp:139-142): header counts, box bounds (+ tilt),
    Masses and Atoms sections.  `atom_style` comes from the `Atoms # style` comment when not given (VIAMD asks the user when it
    cannot tell, src/loader.cpp:80-88).  Returns (coords float32 [1, 3, N] sorted by atom id, dict(type, mol, mass, id), cell)."""
    with open(path) as f:
        lines = [ln.rstrip("\n") for ln in f]
    lo, hi, tilt = [0.0] * 3, [0.0] * 3, (0.0, 0.0, 0.0)
    natoms, masses, atoms, section = None, {}, [], None
    for ln in lines[1:]:
        body = ln.split("#")[0].strip()
        if not body:
            continue
        t = body.split()
        if body.endswith("atoms") and len(t) == 2:
            natoms = int(t[0])
        elif len(t) == 4 and t[2].endswith("lo") and t[3].endswith("hi"):
            k = "xyz".index(t[2][0])
            lo[k], hi[k] = float(t[0]), float(t[1])
        elif len(t) == 6 and t[3:] == ["xy", "xz", "yz"]:
            tilt = (float(t[0]), float(t[1]), float(t[2]))
        elif t[0] in ("Masses", "Atoms", "Velocities", "Bonds", "Angles", "Dihedrals", "Impropers", "Pair", "PairIJ", "Bond", "Angle",
                      "Dihedral", "Improper", "Ellipsoids", "Lines", "Triangles", "Bodies") and not t[0][0].isdigit():
            section = t[0]
            if section == "Atoms" and atom_style is None and "#" in ln:
                atom_style = ln.split("#")[1].split()[0]
        elif section == "Masses" and len(t) >= 2:
            masses[int(t[0])] = float(t[1])
        elif section == "Atoms":
            atoms.append(t)
    if atom_style is None:
        raise ValueError(f"{path}: cannot determine the LAMMPS atom style (no `Atoms # style` comment): pass atom_style")
    if atom_style not in LAMMPS_ATOM_STYLES:
        raise ValueError(f"{path}: unsupported atom style '{atom_style}'")
    st = LAMMPS_ATOM_STYLES[atom_style]
    if not atoms or (natoms is not None and len(atoms) != natoms):
        raise ValueError(f"{path}: the Atoms section has {len(atoms)} lines, the header says {natoms}")
    atoms.sort(key=lambda t: int(t[0]))
    xyz = np.array([[float(t[st["x"] + k]) for k in range(3)] for t in atoms], np.float32).T
    types = np.array([int(t[st["type"]]) for t in atoms])
    meta = dict(id=np.array([int(t[0]) for t in atoms]), type=types,
                mol=np.array([int(t[st["mol"]]) for t in atoms]) if st["mol"] is not None else np.zeros(len(atoms), np.int64),
                mass=np.array([masses.get(int(tp), 0.0) for tp in types], np.float32))
    cell = make_unitcell((hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]), tilt=tilt)
    return xyz[None], meta, cell
