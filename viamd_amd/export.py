"""Property export as VIAMD writes it: XVG / CSV tables for temporal and distribution properties
(/root/reference/src/main.cpp:5640-5716, 6017-6046) and Gaussian cube files for volumes (export_cube, src/main.cpp:5718-5830)."""
import numpy as np

ANGSTROM_TO_BOHR = 1.0 / 0.529177210903      # src/main.cpp:5766


def write_cube(path, volume, dim, extent, atoms=None):
    """volume: float array [dim^3], x fastest (values[z*d*d + y*d + x]); extent = half edge in Angstrom (vis.sdf.extent).
    atoms: optional list of (atomic_number, x, y, z) in the volume's reference frame, Angstrom.
    Layout follows export_cube: Bohr units, origin -extent, voxel = 2*extent/dim, OUTER LOOP X, MIDDLE Y, INNER Z, six
    values per line in %12.6E."""
    v = np.asarray(volume, np.float32).reshape(dim, dim, dim)          # [z][y][x]
    atoms = atoms or []
    ext = 2.0 * extent * ANGSTROM_TO_BOHR
    vox = ext / dim
    half = 0.5 * ext
    with open(path, "w") as f:
        f.write("EXPORTED DENSITY VOLUME FROM VIAMD, UNITS IN BOHR\n")
        f.write("OUTER LOOP: X, MIDDLE LOOP: Y, INNER LOOP: Z\n")
        f.write("%5i %12.6f %12.6f %12.6f\n" % (-len(atoms), -half, -half, -half))
        f.write("%5i %12.6f %12.6f %12.6f\n" % (dim, vox, 0.0, 0.0))
        f.write("%5i %12.6f %12.6f %12.6f\n" % (dim, 0.0, vox, 0.0))
        f.write("%5i %12.6f %12.6f %12.6f\n" % (dim, 0.0, 0.0, vox))
        for anum, x, y, z in atoms:
            f.write("%5i %12.6f %12.6f %12.6f %12.6f\n" % (anum, float(anum), x * ANGSTROM_TO_BOHR, y * ANGSTROM_TO_BOHR, z * ANGSTROM_TO_BOHR))
        f.write("%5i %5i\n" % (1, 1))
        count = 0
        xyz = np.transpose(v, (2, 1, 0))                                # [x][y][z]: z innermost, as the reference loops
        out = []
        for val in xyz.reshape(-1):
            out.append(" %12.6E" % val)
            count += 1
            if count % 6 == 0:
                out.append("\n")
        f.write("".join(out))


def read_cube(path):
    """Inverse of write_cube (for tests / diffing against files exported by a real VIAMD). Returns dict."""
    with open(path) as f:
        lines = f.read().split("\n")
    nat, ox, oy, oz = lines[2].split()
    nat = int(nat)
    dims, vox = [], []
    for k in range(3):
        t = lines[3 + k].split()
        dims.append(int(t[0])); vox.append(float(t[1 + k]))
    n = abs(nat)
    atoms = [tuple(float(x) for x in lines[6 + i].split()) for i in range(n)]
    body = 6 + n + (1 if nat < 0 else 0)
    vals = np.array(" ".join(lines[body:]).split(), np.float64)
    vol = vals.reshape(dims[0], dims[1], dims[2]).transpose(2, 1, 0)    # back to [z][y][x]
    return {"origin": (float(ox), float(oy), float(oz)), "dim": dims, "voxel": vox, "atoms": atoms, "volume": vol.reshape(-1)}


def _columns(x_label, x, series):
    cols = [np.asarray(x, np.float64)] + [np.asarray(s, np.float64) for _, s in series]
    return [x_label] + [n for n, _ in series], np.stack(cols, axis=1)


def write_csv(path, x_label, x, series):
    """series: list of (label, values)."""
    head, tab = _columns(x_label, x, series)
    with open(path, "w") as f:
        f.write(",".join(head) + "\n")
        for row in tab:
            f.write(",".join("%g" % v for v in row) + "\n")


def write_xvg(path, title, x_label, y_label, x, series):
    head, tab = _columns(x_label, x, series)
    with open(path, "w") as f:
        f.write("# This file was created by VIAMD (viamd_amd backend)\n")
        f.write('@    title "%s"\n@    xaxis  label "%s"\n@    yaxis  label "%s"\n@TYPE xy\n' % (title, x_label, y_label))
        for i, n in enumerate(head[1:]):
            f.write('@ s%d legend "%s"\n' % (i, n))
        for row in tab:
            f.write(" ".join("%12.6f" % v for v in row) + "\n")


def distribution_table(prop, num_bins=128, lib=None):
    """What VIAMD exports for a distribution property: the display histogram g = sum(values)/sum(weights) per bin over
    [min_range[0], max_range[0]] (src/main.cpp:1515-1525)."""
    from .eval import downsample_histogram
    g = downsample_histogram(prop.values, prop.weights, num_bins, lib=lib)
    lo, hi = prop.min_range[0], prop.max_range[0]
    x = lo + (np.arange(num_bins) + 0.5) * (hi - lo) / num_bins
    return x, g
