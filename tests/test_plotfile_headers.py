"""The plotfile writer's text side against the reference's own files (CPU, no device): rh_plotfile_write_headers, fed with the
box list, data-file offsets and per-box minima / maxima that the gold plotfile records, must reproduce the gold `Header`,
`Level_0/Cell_H` and `Level_0/Nu_nd_H` BYTE FOR BYTE (format of REMORA_Plotfile.cpp:520-605 and AMReX's VisMF header)."""
import ctypes as C
import os
import re

import pytest

from remora_amd import cases, lib as L
from remora_amd.model import make_params
from plotfile import GOLD

CASES = ["Upwelling", "Seamount", "DoubleGyre", "Advection", "DoublyPeriodic", "DoublyPeriodic_bathy", "Upwelling_GLS", "Channel_Test"]


def _parse_vismf(path):
    txt = open(path).read()
    boxes = [tuple(int(x) for x in m) for m in re.findall(r"\(\((-?\d+),(-?\d+),(-?\d+)\) \((-?\d+),(-?\d+),(-?\d+)\) \(\d,\d,\d\)\)", txt)]
    fabs = [(int(i), int(o)) for i, o in re.findall(r"FabOnDisk: \w+_D_(\d+) (\d+)", txt)]
    blocks = re.findall(r"\d+,\d+\n((?:[-+0-9.eE,]+\n)+)", txt)
    rows = lambda b: [[float(x) for x in r.strip(",").split(",")] for r in b.strip().split("\n")]
    return boxes, fabs, rows(blocks[0]), rows(blocks[1])


@pytest.mark.parametrize("case", CASES)
def test_headers_reproduce_gold_text(case, tmp_path):
    g = os.path.join(GOLD, case)
    hdr = open(os.path.join(g, "Header")).read().split("\n")
    ncomp = int(hdr[1]); time = float(hdr[3 + ncomp]); step = int(hdr[9 + ncomp].split()[0])
    cb, cf, cmin, cmax = _parse_vismf(os.path.join(g, "Level_0", "Cell_H"))
    nb, nf, nmin, nmax = _parse_vismf(os.path.join(g, "Level_0", "Nu_nd_H"))
    n = len(cb)
    B = (L.PlotBox * n)()
    for q in range(n):
        for d in range(3):
            B[q].box.lo[d] = cb[q][d]; B[q].box.hi[d] = cb[q][3 + d]
        B[q].file_index = cf[q][0]; B[q].cell_offset = cf[q][1]; B[q].nd_offset = nf[q][1]
        assert nf[q][0] == cf[q][0]
        for c in range(ncomp):
            B[q].cell_min[c] = cmin[q][c]; B[q].cell_max[c] = cmax[q][c]
        for c in range(3):
            B[q].nd_min[c] = nmin[q][c]; B[q].nd_max[c] = nmax[q][c]
    p = make_params(cases.get(case))
    out = str(tmp_path / "plt")
    L.check(L.load().rh_plotfile_write_headers(C.byref(p), out.encode(), int(ncomp == 6), step, time, n, B))
    for rel in ("Header", os.path.join("Level_0", "Cell_H"), os.path.join("Level_0", "Nu_nd_H")):
        mine = open(os.path.join(out, rel)).read(); gold = open(os.path.join(g, rel)).read()
        assert mine == gold, f"{case} {rel}:\n--- mine ---\n{mine[:1500]}\n--- gold ---\n{gold[:1500]}"
