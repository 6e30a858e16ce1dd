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


# ---------------------------------------------------------------------------------------------------------------------------
# The starred leaves through their C-ABI entry points against the oracle's restatement of the same reference lambda
# (oracle/leaves.c, bound in oracle/oracle_py.py), on seeded random operands (seed 12345, the ranges of SURVEY.md section 8d:
# Hz in [1, 20] m, |u| <= 0.5, T in [5, 25], Akv in [1e-5, 1e-2]); transports and W of both signs; boxes that do not fill their
# last tile in either direction (150 = 2 x 64 + 22 columns, 21 = 2 x 8 + 5 rows).  Bit-exact.
from oracle import oracle_py as O          # noqa: E402

NX, NY = 150, 21


class Arr:
    """one field: host array [comp, k, j, i] with its index origin, the oracle's Fab over it and a device copy with its rh_array4"""

    def __init__(self, host, lo):
        self.h = np.ascontiguousarray(host, dtype=np.float64)
        self.lo = tuple(lo)
        self.d = torch.tensor(self.h, device="cuda:0")
        self.h0 = self.h.copy()

    def fab(self):
        return O.ofab(self.h, self.lo)

    def a4(self):
        return _a4(self.d, self.lo)

    def same(self):
        return np.array_equal(self.d.cpu().numpy(), self.h)

    def changed(self):
        return not np.array_equal(self.h, self.h0)


def _rand(rng, lo_v, hi_v, nc, nk, lo, hi):
    """uniform field on the index box lo..hi (2-D in i, j), nk levels"""
    return Arr(rng.uniform(lo_v, hi_v, (nc, nk, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1)), (lo[0], lo[1], 0))


def _ctx(nz, periodic, land=False, **flags):
    """flags: SolverChoice switches the leaf reads (tracer_c4, uv_c2, flat_bathymetry); the pinned cases all run the defaults, so
    the other arms are PARITY UNPINNED (no gold file) and stand on the oracle's source reading"""
    cfg = cases.get("Seamount" if periodic else "DoubleGyre")
    cfg["ncell"] = (NX, NY, nz)
    cfg.update(flags)
    if land:
        cfg["masks_all_one"] = 0            # the kernels read the mask arrays they are given
    return cfg, REMORA(cfg)


def _mask(land, shape2d, lo, values=(0.0, 1.0), seed=777):
    """a 2-D mask array: ones, or (land) seeded random picks from `values` -- about a third of the points dry"""
    if not land:
        return Arr(np.ones((1, 1) + shape2d), lo)
    rng = np.random.default_rng(seed)
    pick = rng.choice(len(values) + 1, size=shape2d)          # one extra slot for "wet" so that water dominates
    m = np.where(pick >= len(values), 1.0, np.asarray(values)[np.minimum(pick, len(values) - 1)])
    return Arr(m[None, None], lo)


def _metrics(rng, g):
    return (_rand(rng, 1e-3, 2e-3, 1, 1, (-g, -g - 1), (NX - 1 + g, NY + g)), _rand(rng, 1e-3, 2e-3, 1, 1, (-g - 1, -g), (NX + g, NY - 1 + g)))


@pytest.mark.parametrize("c4,flat", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("land", [False, True])
@pytest.mark.parametrize("periodic", [True, False])
def test_rhs_t_3d_leaf(periodic, land, c4, flat):
    """REMORA::rhs_t_3d, REMORA_rhs_t_3d.cpp:27-335: upstream3 / centered4 (:112-141,148-177,209-240,244-275), each with the
    flat-bathymetry form of the correction term"""
    nz = 7; N = nz - 1
    cfg, rem = _ctx(nz, periodic, land, tracer_c4=c4, flat_bathymetry=flat)
    rng = np.random.default_rng(12345)
    g = 3
    t = _rand(rng, 5.0, 25.0, 1, nz, (-g, -g), (NX - 1 + g, NY - 1 + g)); ss = _rand(rng, 5.0, 25.0, 1, nz, (-g, -g), (NX - 1 + g, NY - 1 + g))
    Huon = _rand(rng, -50.0, 50.0, 1, nz, (-g, -g), (NX + g, NY - 1 + g)); Hvom = _rand(rng, -50.0, 50.0, 1, nz, (-g, -g), (NX - 1 + g, NY + g))
    Hz = _rand(rng, 1.0, 20.0, 1, nz, (-g, -g), (NX - 1 + g, NY - 1 + g)); W = _rand(rng, -5.0, 5.0, 1, nz + 1, (-g, -g), (NX - 1 + g, NY - 1 + g))
    pm, pn = _metrics(rng, g)
    msku = _mask(land, (NY + 2 * g, NX + 1 + 2 * g), (-g, -g, 0), seed=1); mskv = _mask(land, (NY + 1 + 2 * g, NX + 2 * g), (-g, -g, 0), seed=2)
    FC = Arr(np.zeros((1, nz + 1, NY + 2 * g, NX + 2 * g)), (-g, -g, 0))
    dt = 300.0
    bx = ((0, 0, 0), (NX - 1, NY - 1, N))
    O.leaf("o_rhs_t_3d")(O.geom(cfg), O.obox(*bx), O.obox((-g, -g, 0), (NX - 1 + g, NY - 1 + g, N)), t.fab(), ss.fab(), Huon.fab(), Hvom.fab(), Hz.fab(),
                         pn.fab(), pm.fab(), W.fab(), FC.fab(), msku.fab(), mskv.fab(), 0, 0, N, dt)
    L.check(rem.lib.rh_rhs_t_3d(rem.ctx, L.box(*bx), t.a4(), ss.a4(), Huon.a4(), Hvom.a4(), Hz.a4(), pn.a4(), pm.a4(), W.a4(), msku.a4(), mskv.a4(), N, dt))
    rem.sync()
    assert t.changed() and t.same()
    rem.close()


@pytest.mark.parametrize("c2", [0, 1])
@pytest.mark.parametrize("periodic", [True, False])
def test_rhs_uv_3d_leaf(periodic, c2):
    """REMORA::coriolis + REMORA::rhs_uv_3d (REMORA_coriolis.cpp:10-40, REMORA_rhs_uv_3d.cpp:31-385): domain-edge curvature branches
    with walls, none with periodic sides; upstream3 and centered2 (:148-157,302-311) momentum fluxes"""
    nz = 7; N = nz - 1
    cfg, rem = _ctx(nz, periodic, uv_c2=c2)
    rng = np.random.default_rng(12345)
    g = 3
    cb = ((-g, -g), (NX - 1 + g, NY - 1 + g)); ub = ((-g, -g), (NX + g, NY - 1 + g)); vb = ((-g, -g), (NX - 1 + g, NY + g))
    uold = _rand(rng, -0.5, 0.5, 1, nz, *ub); vold = _rand(rng, -0.5, 0.5, 1, nz, *vb)
    ru = _rand(rng, -1.0, 1.0, 2, nz, *ub); rv = _rand(rng, -1.0, 1.0, 2, nz, *vb)
    rufrc = _rand(rng, -1.0, 1.0, 2, 1, *ub); rvfrc = _rand(rng, -1.0, 1.0, 2, 1, *vb)
    sustr = _rand(rng, -1e-4, 1e-4, 1, 1, *ub); svstr = _rand(rng, -1e-4, 1e-4, 1, 1, *vb)
    bustr = _rand(rng, -1e-4, 1e-4, 1, 1, *ub); bvstr = _rand(rng, -1e-4, 1e-4, 1, 1, *vb)
    Huon = _rand(rng, -50.0, 50.0, 1, nz, *ub); Hvom = _rand(rng, -50.0, 50.0, 1, nz, *vb)
    W = _rand(rng, -5.0, 5.0, 1, nz + 1, *cb); Hz = _rand(rng, 1.0, 20.0, 1, nz, *cb)
    pm = _rand(rng, 1e-3, 2e-3, 1, 1, *cb); pn = _rand(rng, 1e-3, 2e-3, 1, 1, *cb); fcor = _rand(rng, -1e-4, 1e-4, 1, 1, *cb)
    fomn = Arr(fcor.h * (1.0 / (pm.h * pn.h)), fcor.lo)                       # REMORA_setup_step.cpp:233-239
    FC = Arr(np.zeros((1, nz + 1, NY + 1 + 2 * g, NX + 1 + 2 * g)), (-g, -g, 0))
    xbx = ((0, 0, 0), (NX, NY - 1, N)); ybx = ((0, 0, 0), (NX - 1, NY, N))
    use_cor = 1
    O.leaf("o_coriolis")(O.obox(*xbx), O.obox(*ybx), uold.fab(), vold.fab(), ru.fab(), rv.fab(), Hz.fab(), fomn.fab(), 0, 0)
    O.leaf("o_rhs_uv_3d")(O.geom(cfg), O.obox(*xbx), O.obox(*ybx), uold.fab(), vold.fab(), ru.fab(), rv.fab(), rufrc.fab(), rvfrc.fab(), sustr.fab(), svstr.fab(),
                          bustr.fab(), bvstr.fab(), Huon.fab(), Hvom.fab(), pm.fab(), pn.fab(), W.fab(), FC.fab(), 0, N)
    L.check(rem.lib.rh_rhs_uv_3d(rem.ctx, L.box(*xbx), L.box(*ybx), uold.a4(), vold.a4(), ru.a4(), rv.a4(), rufrc.a4(), rvfrc.a4(), sustr.a4(), svstr.a4(),
                                 bustr.a4(), bvstr.a4(), Huon.a4(), Hvom.a4(), pm.a4(), pn.a4(), W.a4(), Hz.a4(), fcor.a4(), use_cor, 0, N))
    rem.sync()
    for a in (ru, rv, rufrc, rvfrc):
        assert a.changed() and a.same()
    rem.close()


@pytest.mark.parametrize("land", [False, True])
def test_uv3dmix_leaf(land):
    """REMORA::uv3dmix, REMORA_uv3dmix.cpp:6-149 (3-D call)"""
    nz = 7; N = nz - 1
    cfg, rem = _ctx(nz, False, land)
    rng = np.random.default_rng(12345)
    g = 3
    cb = ((-g, -g), (NX - 1 + g, NY - 1 + g)); ub = ((-g, -g), (NX + g, NY - 1 + g)); vb = ((-g, -g), (NX - 1 + g, NY + g)); pb = ((-g, -g), (NX + g, NY + g))
    u = _rand(rng, -0.5, 0.5, 1, nz, *ub); v = _rand(rng, -0.5, 0.5, 1, nz, *vb)
    uold = _rand(rng, -0.5, 0.5, 1, nz, *ub); vold = _rand(rng, -0.5, 0.5, 1, nz, *vb)
    rufrc = _rand(rng, -1.0, 1.0, 2, 1, *ub); rvfrc = _rand(rng, -1.0, 1.0, 2, 1, *vb)
    visc2_p = _rand(rng, 1.0, 100.0, 1, 2, *pb); visc2_r = _rand(rng, 1.0, 100.0, 1, 2, *cb)
    Hz = _rand(rng, 1.0, 20.0, 1, nz, *cb)
    pm = _rand(rng, 1e-3, 2e-3, 1, 1, (-g - 1, -g - 1), (NX + g, NY + g)); pn = _rand(rng, 1e-3, 2e-3, 1, 1, (-g - 1, -g - 1), (NX + g, NY + g))
    mskp = _mask(land, (NY + 1 + 2 * g, NX + 1 + 2 * g), (-g, -g, 0), values=(0.0, 2.0), seed=3)      # update_mskp's values: 0, 1, 2
    xbx = ((0, 0, 0), (NX, NY - 1, N)); ybx = ((0, 0, 0), (NX - 1, NY, N))
    dt = 300.0
    O.leaf("o_uv3dmix")(O.obox(*xbx), O.obox(*ybx), u.fab(), v.fab(), uold.fab(), vold.fab(), rufrc.fab(), rvfrc.fab(), visc2_p.fab(), visc2_r.fab(), Hz.fab(),
                        pm.fab(), pn.fab(), mskp.fab(), 0, 0, dt)
    L.check(rem.lib.rh_uv3dmix(rem.ctx, L.box(*xbx), L.box(*ybx), u.a4(), v.a4(), uold.a4(), vold.a4(), rufrc.a4(), rvfrc.a4(), visc2_p.a4(), visc2_r.a4(),
                               Hz.a4(), pm.a4(), pn.a4(), mskp.a4(), 0, 0, dt))
    rem.sync()
    for a in (u, v, rufrc, rvfrc):
        assert a.changed() and a.same()
    rem.close()


@pytest.mark.parametrize("nz", [7, 16])          # 7: LDS-ring solver, 16: register-resident solver
@pytest.mark.parametrize("ioff,joff", [(0, 0), (1, 0), (0, 1)])
def test_vert_visc_3d_leaf(nz, ioff, joff):
    """REMORA::vert_visc_3d, REMORA_vert_visc_3d.cpp:10-107: tracer, u and v forms"""
    N = nz - 1
    cfg, rem = _ctx(nz, True)
    rng = np.random.default_rng(12345)
    g = 2
    hi = (NX - 1 + ioff, NY - 1 + joff)
    fb = ((-g, -g), (hi[0] + g, hi[1] + g))
    phi = _rand(rng, 5.0, 25.0, 1, nz, *fb); Hz = _rand(rng, 1.0, 20.0, 1, nz, *fb); Akv = _rand(rng, 1e-5, 1e-2, 1, nz + 1, *fb)
    scr = [Arr(np.zeros((1, nz + 2, hi[1] + 1 + 2 * g, hi[0] + 1 + 2 * g)), (-g, -g, 0)) for _ in range(6)]       # Hzk, AK, BC, DC, FC, CF
    bx = ((0, 0, 0), (hi[0], hi[1], N))
    dt = 300.0
    O.leaf("o_vert_visc_3d")(O.obox(*bx), ioff, joff, phi.fab(), Hz.fab(), scr[0].fab(), scr[1].fab(), Akv.fab(), scr[2].fab(), scr[3].fab(), scr[4].fab(),
                             scr[5].fab(), 0, N, dt)
    L.check(rem.lib.rh_vert_visc_3d(rem.ctx, L.box(*bx), ioff, joff, phi.a4(), Hz.a4(), Akv.a4(), N, dt))
    rem.sync()
    assert phi.changed() and phi.same()
    rem.close()


@pytest.mark.parametrize("c2", [0, 1])
@pytest.mark.parametrize("periodic", [True, False])
def test_rhs_uv_2d_leaf(periodic, c2):
    """REMORA::rhs_uv_2d, REMORA_rhs_uv_2d.cpp:20-238 (the entry point the fused barotropic kernel never calls); upstream3 and
    centered2 (:130,217)"""
    cfg, rem = _ctx(7, periodic, uv_c2=c2)
    rng = np.random.default_rng(12345)
    g = 3
    ub = ((-g, -g), (NX + g, NY - 1 + g)); vb = ((-g, -g), (NX - 1 + g, NY + g))
    ubar = _rand(rng, -0.5, 0.5, 3, 1, *ub); vbar = _rand(rng, -0.5, 0.5, 3, 1, *vb)
    rhs_u = _rand(rng, -1.0, 1.0, 1, 1, *ub); rhs_v = _rand(rng, -1.0, 1.0, 1, 1, *vb)
    DUon = _rand(rng, -500.0, 500.0, 1, 1, *ub); DVom = _rand(rng, -500.0, 500.0, 1, 1, *vb)
    xbx = ((0, 0, 0), (NX, NY - 1, 0)); ybx = ((0, 0, 0), (NX - 1, NY, 0))
    for krhs in (0, 2):
        O.leaf("o_rhs_uv_2d")(O.geom(cfg), O.obox(*xbx), O.obox(*ybx), ubar.fab(), vbar.fab(), rhs_u.fab(), rhs_v.fab(), DUon.fab(), DVom.fab(), krhs)
        L.check(rem.lib.rh_rhs_uv_2d(rem.ctx, L.box(*xbx), L.box(*ybx), ubar.a4(), vbar.a4(), rhs_u.a4(), rhs_v.a4(), DUon.a4(), DVom.a4(), krhs))
        rem.sync()
        assert rhs_u.changed() and rhs_u.same() and rhs_v.same()
    rem.close()


@pytest.mark.parametrize("periodic", [True, False])
@pytest.mark.parametrize("land", [False, True])
@pytest.mark.parametrize("ioff,joff", [(1, 0), (0, 1)])
def test_update_massflux_3d_leaf(periodic, ioff, joff, land):
    """REMORA::update_massflux_3d, REMORA_update_massflux_3d.cpp:24-122, on the grow-2 box the driver uses (walls: the boundary-face
    velocity correction of :87-101)"""
    nz = 7; N = nz - 1
    cfg, rem = _ctx(nz, periodic, land)
    rng = np.random.default_rng(12345)
    g = 3
    fb = ((-g, -g), (NX - 1 + ioff + g, NY - 1 + joff + g)); cb = ((-g, -g), (NX - 1 + g, NY - 1 + g))
    phi = _rand(rng, -0.5, 0.5, 1, nz, *fb); Hphi = _rand(rng, -50.0, 50.0, 1, nz, *fb); Hz = _rand(rng, 1.0, 20.0, 1, nz, *cb)
    phibar = _rand(rng, -0.5, 0.5, 3, 1, *fb); pmn = _rand(rng, 1e-3, 2e-3, 1, 1, (-g - 1, -g - 1), (NX + g, NY + g))
    a1 = _rand(rng, -500.0, 500.0, 1, 1, *fb); a2 = _rand(rng, -500.0, 500.0, 1, 1, *fb)
    msk = _mask(land, (fb[1][1] + g + 1, fb[1][0] + g + 1), (-g, -g, 0), seed=4)
    DC = Arr(np.zeros((1, nz + 2, fb[1][1] + g + 1, fb[1][0] + g + 1)), (-g, -g, -1)); FC = Arr(np.zeros((1, nz + 1, fb[1][1] + g + 1, fb[1][0] + g + 1)), (-g, -g, 0))
    DC.lo = (-g, -g, -1)
    bx = ((-2, -2, 0), (NX - 1 + 2, NY - 1 + 2, N))
    O.leaf("o_update_massflux_3d")(O.geom(cfg), O.obox(*bx), ioff, joff, phi.fab(), phibar.fab(), Hphi.fab(), Hz.fab(), pmn.fab(), a1.fab(), a2.fab(),
                                   DC.fab(), FC.fab(), msk.fab(), 0)
    L.check(rem.lib.rh_update_massflux_3d(rem.ctx, L.box(*bx), ioff, joff, phi.a4(), phibar.a4(), Hphi.a4(), Hz.a4(), pmn.a4(), a1.a4(), a2.a4(), msk.a4(), N))
    rem.sync()
    assert Hphi.changed() and Hphi.same() and phibar.same() and phi.same()
    if not periodic:
        assert phi.changed()
    rem.close()


@pytest.mark.parametrize("c4,flat", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("land", [False, True])
def test_prestep_t_leaf(land, c4, flat):
    """omega + REMORA::prestep_t_advection + the tracer call of prestep_diffusion (REMORA_prestep_t_advection.cpp:10-361,
    REMORA_prestep_diffusion.cpp:19-165 with lambda = 1) for one tracer, second-and-later-step weights (iic = 2); upstream3,
    centered4 (:222-234,255-266) and the flat-bathymetry centred flux (:147-157, which ignores the scheme)"""
    nz = 7; N = nz - 1
    cfg, rem = _ctx(nz, True, land, tracer_c4=c4, flat_bathymetry=flat)
    rng = np.random.default_rng(12345)
    g = 3
    cb = ((-g, -g), (NX - 1 + g, NY - 1 + g)); ub = ((-g, -g), (NX + g, NY - 1 + g)); vb = ((-g, -g), (NX - 1 + g, NY + g))
    told = _rand(rng, 5.0, 25.0, 1, nz, *cb); tnew = _rand(rng, 5.0, 25.0, 1, nz, *cb); sst = _rand(rng, 0.0, 1.0, 1, nz, *cb)
    cache = Arr(tnew.h.copy(), tnew.lo)
    Hz = _rand(rng, 1.0, 20.0, 1, nz, *cb); Huon = _rand(rng, -50.0, 50.0, 1, nz, *ub); Hvom = _rand(rng, -50.0, 50.0, 1, nz, *vb)
    z_w = Arr(np.cumsum(np.concatenate([np.full((1, 1) + Hz.h.shape[2:], -200.0), Hz.h], axis=1), axis=1), Hz.lo)
    z_r = Arr(0.5 * (z_w.h[:, 1:] + z_w.h[:, :-1]), Hz.lo)
    h = Arr(np.repeat(-z_w.h[:, :1], 2, axis=0), Hz.lo)
    W = Arr(np.zeros((1, nz + 1) + Hz.h.shape[2:]), Hz.lo)
    pm, pn = (_rand(rng, 1e-3, 2e-3, 1, 1, *cb), _rand(rng, 1e-3, 2e-3, 1, 1, *cb))
    msku = _mask(land, (NY + 2 * g, NX + 1 + 2 * g), (-g, -g, 0), seed=5); mskv = _mask(land, (NY + 1 + 2 * g, NX + 2 * g), (-g, -g, 0), seed=6)
    Akt = _rand(rng, 1e-5, 1e-2, 3, nz + 1, *cb); ru = _rand(rng, -1.0, 1.0, 2, nz, *ub)
    zero2 = Arr(np.zeros((1, 1) + Hz.h.shape[2:]), Hz.lo)
    DC = Arr(np.zeros((1, nz + 2) + Hz.h.shape[2:]), (-g, -g, -1)); FC = Arr(np.zeros((1, nz + 2) + Hz.h.shape[2:]), Hz.lo)
    dt = 300.0; iic = 2
    bx = ((0, 0, 0), (NX - 1, NY - 1, N)); gbx = ((-g, -g, -3), (NX - 1 + g, NY - 1 + g, N + 3))
    geo = O.geom(cfg)
    O.leaf("o_prestep_t_advection")(geo, O.obox(*bx), O.obox(*gbx), told.fab(), cache.fab(), Hz.fab(), Huon.fab(), Hvom.fab(), W.fab(), DC.fab(), FC.fab(),
                                    sst.fab(), z_w.fab(), h.fab(), pm.fab(), pn.fab(), msku.fab(), mskv.fab(), iic, 0, 0, N, dt)
    O.leaf("o_prestep_diffusion")(geo, O.obox(*bx), O.obox(*gbx), 0, 0, tnew.fab(), told.fab(), ru.fab(), Hz.fab(), Akt.fab(), DC.fab(), FC.fab(),
                                  zero2.fab(), zero2.fab(), z_r.fab(), pm.fab(), pn.fab(), iic, iic, 0, 0, 0, N, 1.0, dt)
    g1 = ((-1, -1, 0), (NX, NY, N))
    L.check(rem.lib.rh_omega(rem.ctx, L.box(*g1), W.a4(), Huon.a4(), Hvom.a4(), z_w.a4(), h.a4(), N))
    L.check(rem.lib.rh_prestep_t(rem.ctx, L.box(*bx), told.a4(), tnew.a4(), sst.a4(), Hz.a4(), Huon.a4(), Hvom.a4(), W.a4(), pm.a4(), pn.a4(),
                                 msku.a4(), mskv.a4(), iic, 0, N, dt))
    rem.sync()
    assert W.changed() and W.same()
    assert sst.changed() and sst.same() and tnew.changed() and tnew.same()
    rem.close()
