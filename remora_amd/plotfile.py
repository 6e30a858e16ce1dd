"""Reader of AMReX-native plotfiles (the format REMORA writes, IO/REMORA_Plotfile.cpp:141-605, and rh_write_plotfile reproduces) and
the norm-wise comparison of amrex_fcompare as the reference's regression tests run it (Tests/CTestList.cmake:39:
`amrex_fcompare -a -r 1e-11 --abs_tol 1e-11 gold plt00010`).  Host-only (numpy); tools/rh_fcompare.py is the command-line front end.

Layout read here: `Header` = component names, domain box; `Level_0/Cell_H` (cell data) or `Level_0/Nu_nd_H` (the
nodal mesh-height block) = FabOnDisk records (file, offset); each record = one ASCII FAB header line followed by raw little-endian
doubles, i fastest, then j, k, component.  Boxes are stitched into the level-0 domain, so two plotfiles written with different box
layouts (one box, nine boxes, one box per rank) compare like fcompare's -a / --allow_diff_grids.
"""
import os
import re

import numpy as np


def read_header(d):
    """(names, lo, hi, time) of a plotfile directory"""
    with open(os.path.join(d, "Header")) as f:
        lines = f.read().split("\n")
    ncomp = int(lines[1])
    names = lines[2:2 + ncomp]
    time = float(lines[2 + ncomp + 1])
    dom = re.findall(r"\(\((-?\d+),(-?\d+),(-?\d+)\) \((-?\d+),(-?\d+),(-?\d+)\)", "\n".join(lines))[0]
    return names, tuple(int(x) for x in dom[:3]), tuple(int(x) for x in dom[3:]), time


def read_plotfile_dir(d, nodal=False):
    """(names, data[comp, k, j, i]) of the plotfile directory d; nodal=True reads the Nu_nd block."""
    names, lo, hi, _ = read_header(d)
    ncomp = len(names)
    if nodal:
        hi = tuple(h + 1 for h in hi); ncomp = 3; names = ["amrexvec_nu_x", "amrexvec_nu_y", "amrexvec_nu_z"]
    shape = (hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1)
    out = np.full((ncomp,) + shape, np.nan)
    with open(os.path.join(d, "Level_0", "Nu_nd_H" if nodal else "Cell_H")) as f:
        ch = f.read()
    for fname, off in re.findall(r"FabOnDisk: (\S+) (\d+)", ch):
        with open(os.path.join(d, "Level_0", fname), "rb") as f:
            f.seek(int(off))
            hdr = f.readline().decode()
            m = re.search(r"\(\((-?\d+),(-?\d+),(-?\d+)\) \((-?\d+),(-?\d+),(-?\d+)\) \(\d+,\d+,\d+\)\) (\d+)", hdr)
            blo = [int(m.group(i)) for i in (1, 2, 3)]; bhi = [int(m.group(i)) for i in (4, 5, 6)]
            nc = int(m.group(7))
            n = [bhi[i] - blo[i] + 1 for i in range(3)]
            data = np.frombuffer(f.read(8 * nc * n[0] * n[1] * n[2]), dtype="<f8").reshape(nc, n[2], n[1], n[0])
            out[:, blo[2]-lo[2]:bhi[2]-lo[2]+1, blo[1]-lo[1]:bhi[1]-lo[1]+1, blo[0]-lo[0]:bhi[0]-lo[0]+1] = data
    if np.isnan(out).any():
        raise ValueError(f"{d}: the boxes on disk do not cover the level-0 domain")
    return names, out


def fcompare(gold, mine, rtol=1e-11, atol=1e-11):
    """Per-component (abs_err, rel_err, ok) with fcompare's norm-wise criterion: the max-norm of the difference, absolute and
    relative to the max-norm of the first file's component; a component passes when either is within its tolerance."""
    res = []
    for c in range(gold.shape[0]):
        aerr = float(np.max(np.abs(gold[c] - mine[c])))
        gmax = float(np.max(np.abs(gold[c])))
        rerr = aerr / gmax if gmax > 0 else (0.0 if aerr == 0 else np.inf)
        res.append((aerr, rerr, bool(aerr <= atol or rerr <= rtol)))
    return res


def fcompare_dirs(dir_a, dir_b, rtol=0.0, atol=0.0, nodal=False):
    """Compare two plotfile directories: returns (ok, rows, problems); rows = (name, abs_err, rel_err, ok) per variable of dir_a.
    Variables are matched by name (as fcompare does); a variable missing from dir_b, or different domains, are problems."""
    na, loa, hia, _ = read_header(dir_a)
    nb, lob, hib, _ = read_header(dir_b)
    problems = []
    if (loa, hia) != (lob, hib):
        return False, [], [f"domains differ: {loa}-{hia} vs {lob}-{hib}"]
    names_a, a = read_plotfile_dir(dir_a, nodal)
    names_b, b = read_plotfile_dir(dir_b, nodal)
    rows = []
    for ia, name in enumerate(names_a):
        if name not in names_b:
            problems.append(f"variable {name} is not in {dir_b}")
            continue
        aerr, rerr, ok = fcompare(a[ia:ia + 1], b[names_b.index(name):names_b.index(name) + 1], rtol, atol)[0]
        rows.append((name, aerr, rerr, ok))
    return (not problems) and all(r[3] for r in rows), rows, problems
