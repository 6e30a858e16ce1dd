"""Leaf entry points that the fused drivers never call on their own: checked against a numpy restatement of the reference lambda
on seeded random inputs (numpy evaluates the same IEEE operations in the same order, no FMA: bit-exact)."""
import ctypes as C

import numpy as np
import pytest
import torch

from remora_amd import cases, lib as L
from remora_amd.model import REMORA

pytestmark = pytest.mark.gpu


def _a4(t, lo):
    """rh_array4 over a torch tensor [comp, k, j, i] whose first element has index lo"""
    a = L.Array4()
    a.p = t.data_ptr()
    nc, nk, nj, ni = t.shape
    for d, (l, n) in enumerate(zip(lo, (ni, nj, nk))):
        a.lo[d] = l; a.hi[d] = l + n - 1
    a.ncomp = nc
    return a


def test_coriolis_leaf():
    """REMORA::coriolis (REMORA_coriolis.cpp:10-40) on a 37 x 21 x 5 box with ghost cells."""
    rem = REMORA(cases.get("Seamount"))          # any context: the leaf only uses its stream
    rng = np.random.default_rng(12345)
    nx, ny, nz, ng = 37, 21, 5, 2
    lo = (-ng, -ng, 0)
    shp3 = (nz, ny + 2 * ng + 1, nx + 2 * ng + 1)
    u = rng.uniform(-0.5, 0.5, (1,) + shp3); v = rng.uniform(-0.5, 0.5, (1,) + shp3)
    Hz = rng.uniform(1.0, 20.0, (1,) + shp3); fomn = rng.uniform(-1e3, 1e3, (1, 1) + shp3[1:])
    ru = rng.uniform(-1.0, 1.0, (2,) + shp3); rv = rng.uniform(-1.0, 1.0, (2,) + shp3)
    dev = {k: torch.tensor(x, device="cuda:0") for k, x in dict(u=u, v=v, Hz=Hz, fomn=fomn, ru=ru, rv=rv).items()}
    xbx = L.box((0, 0, 0), (nx, ny - 1, nz - 1)); ybx = L.box((0, 0, 0), (nx - 1, ny, nz - 1))
    nr = 1
    L.check(rem.lib.rh_coriolis(rem.ctx, xbx, ybx, _a4(dev["u"], lo), _a4(dev["v"], lo), _a4(dev["ru"], lo), _a4(dev["rv"], lo),
                                _a4(dev["Hz"], lo), _a4(dev["fomn"], (lo[0], lo[1], 0)), 0, nr))
    rem.sync()

    def at(a, di, dj, i0, i1, j0, j1):             # a[..., j0+dj : j1+dj+1, i0+di : i1+di+1] in global indices
        return a[..., j0 + dj - lo[1]:j1 + dj - lo[1] + 1, i0 + di - lo[0]:i1 + di - lo[0] + 1]
    exp_ru = ru.copy(); exp_rv = rv.copy()
    i0, i1, j0, j1 = 0, nx, 0, ny - 1
    UFx_i = 0.5 * at(Hz[0], 0, 0, i0, i1, j0, j1) * at(fomn[0], 0, 0, i0, i1, j0, j1) * (at(v[0], 0, 0, i0, i1, j0, j1) + at(v[0], 0, 1, i0, i1, j0, j1))
    UFx_im1 = 0.5 * at(Hz[0], -1, 0, i0, i1, j0, j1) * at(fomn[0], -1, 0, i0, i1, j0, j1) * (at(v[0], -1, 0, i0, i1, j0, j1) + at(v[0], -1, 1, i0, i1, j0, j1))
    at(exp_ru[nr], 0, 0, i0, i1, j0, j1)[...] += 0.5 * (UFx_i + UFx_im1)
    i0, i1, j0, j1 = 0, nx - 1, 0, ny
    VFe_j = 0.5 * at(Hz[0], 0, 0, i0, i1, j0, j1) * at(fomn[0], 0, 0, i0, i1, j0, j1) * (at(u[0], 0, 0, i0, i1, j0, j1) + at(u[0], 1, 0, i0, i1, j0, j1))
    VFe_jm1 = 0.5 * at(Hz[0], 0, -1, i0, i1, j0, j1) * at(fomn[0], 0, -1, i0, i1, j0, j1) * (at(u[0], 0, -1, i0, i1, j0, j1) + at(u[0], 1, -1, i0, i1, j0, j1))
    at(exp_rv[nr], 0, 0, i0, i1, j0, j1)[...] -= 0.5 * (VFe_j + VFe_jm1)
    assert np.array_equal(dev["ru"].cpu().numpy(), exp_ru)
    assert np.array_equal(dev["rv"].cpu().numpy(), exp_rv)
    rem.close()
