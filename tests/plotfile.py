"""Reader for the AMReX native plotfile FABs kept under tests/golden/ (format: SURVEY.md section 4),
and the norm-wise comparison amrex_fcompare applies (Tests/CTestList.cmake:39: -r 1e-11 --abs_tol 1e-11)."""
import os
import re
import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_plotfile(case, root=None, nodal=False):
    """Cell data [comp, k, j, i] of a plotfile directory (default: tests/golden/<case>); nodal=True reads the Nu_nd block."""
    d = os.path.join(GOLD if root is None else root, case)
    with open(os.path.join(d, "Header")) as f:
        lines = f.read().split("\n")
    ncomp = int(lines[1])
    names = lines[2:2 + ncomp]
    dom = re.findall(r"\(\((-?\d+),(-?\d+),(-?\d+)\) \((-?\d+),(-?\d+),(-?\d+)\)", "\n".join(lines))[0]
    lo = tuple(int(x) for x in dom[:3]); hi = tuple(int(x) for x in dom[3:])
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
    assert not np.isnan(out).any()
    return names, out


def fcompare(gold, mine, rtol=1e-11, atol=1e-11):
    """Per-component (abs_err, rel_err, ok) with fcompare's norm-wise criterion."""
    res = []
    for c in range(gold.shape[0]):
        aerr = float(np.max(np.abs(gold[c] - mine[c])))
        gmax = float(np.max(np.abs(gold[c])))
        rerr = aerr / gmax if gmax > 0 else (0.0 if aerr == 0 else np.inf)
        res.append((aerr, rerr, bool(aerr <= atol or rerr <= rtol)))
    return res
