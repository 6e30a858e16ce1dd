"""CPU checks of the oracle's open-boundary restatement (oracle/bc.c) against expectations derived by hand from the reference's
source: the BC table REMORA::init_bcs builds for Exec/Channel_Test/inputs_orlanski and for the pinned SlipWall cases
(Initialization/REMORA_init_bcs.cpp:201-679), and properties of the radiation formula (REMORA_BoundaryConditions_cons.cpp:75-87).
PARITY UNPINNED against reference output: no gold file covers open sides."""
import ctypes as C

import numpy as np

from remora_amd import cases
from oracle import oracle_py as O

CONS, XVEL, YVEL, ZVEL, UBAR, VBAR, ZETA, TKE, FXP, FX, U2D, V2D = 0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13
FOEXTRAP, EXT_DIR, ORL, INT_DIR = 2, 3, 7, 0


def test_table_of_inputs_orlanski():
    g = O.geom(cases.get("Channel_Test_orlanski"))
    for comp in (CONS, CONS + 1, CONS + 2, XVEL, YVEL):          # radiating on the four lateral faces, SlipWall (foextrap) in z
        assert O.get_bc(g, comp) == [ORL, ORL, FOEXTRAP, ORL, ORL, FOEXTRAP], comp
    for comp in (ZVEL, UBAR, VBAR, ZETA, TKE, FXP, FX, U2D, V2D):  # outflow = foextrap; zvel always (:311-316)
        assert O.get_bc(g, comp) == [FOEXTRAP] * 6, comp


def test_table_of_the_pinned_cases():
    g = O.geom(cases.get("DoubleGyre"))                          # four SlipWall sides: only the normal component is ext_dir (:257-271)
    assert O.get_bc(g, XVEL) == [EXT_DIR, FOEXTRAP, FOEXTRAP] * 2
    assert O.get_bc(g, YVEL) == [FOEXTRAP, EXT_DIR, FOEXTRAP] * 2
    assert O.get_bc(g, UBAR) == [EXT_DIR, FOEXTRAP, FOEXTRAP] * 2 and O.get_bc(g, U2D) == [EXT_DIR, FOEXTRAP, FOEXTRAP] * 2
    assert O.get_bc(g, VBAR) == [FOEXTRAP, EXT_DIR, FOEXTRAP] * 2
    assert O.get_bc(g, CONS) == [FOEXTRAP] * 6 and O.get_bc(g, ZETA) == [FOEXTRAP] * 6
    g = O.geom(cases.get("Upwelling"))                           # x periodic, y SlipWall
    assert O.get_bc(g, XVEL) == [INT_DIR, FOEXTRAP, FOEXTRAP] * 2
    assert O.get_bc(g, YVEL) == [INT_DIR, EXT_DIR, FOEXTRAP] * 2
    assert O.get_bc(g, CONS) == [INT_DIR, FOEXTRAP, FOEXTRAP] * 2


def _cons_case(nx=9, ny=8, nz=2):
    cfg = cases.get("Channel_Test_orlanski")
    cfg["ncell"] = (nx, ny, nz)
    return cfg, O.geom(cfg)


def test_radiation_of_a_steady_field_keeps_the_ghost_value():
    """dTdt = 0 (calc equals the new field in the interior): Cx = 0 and the ghost cell takes calc's own ghost value (cff x / cff, to
    rounding), on all four sides; the far ghost lines copy the first one (:193-252) and the corners average (:287-337)."""
    cfg, g = _cons_case()
    nx, ny, nz = cfg["ncell"]
    rng = np.random.default_rng(3)
    ng = 3
    calc = rng.uniform(1.0, 2.0, (3, nz, ny + 2 * ng, nx + 2 * ng))
    dest = calc.copy()
    dest[:, :, :ng, :] = -9.0; dest[:, :, -ng:, :] = -9.0; dest[:, :, :, :ng] = -9.0; dest[:, :, :, -ng:] = -9.0
    one = np.ones((1, 1, ny + 2 * ng + 1, nx + 2 * ng + 1))
    mf = O.omf(dest, (-ng, -ng, 0), (0, 0, 0), (ng, ng, 0))
    O.leaf("o_physbc_calc")(C.byref(g), C.byref(mf), O.ofab(one, (-ng, -ng, 0)), 0, 3, CONS, 0, O.ofab(calc, (-ng, -ng, 0)),
                            O.ofab(one, (-ng, -ng, 0)), O.ofab(one, (-ng, -ng, 0)))
    J = slice(ng, ng + ny); I = slice(ng, ng + nx)
    for q in range(ng):
        np.testing.assert_allclose(dest[:, :, J, q], calc[:, :, J, ng - 1], rtol=4e-16)
        np.testing.assert_allclose(dest[:, :, J, ng + nx + q], calc[:, :, J, ng + nx], rtol=4e-16)
        np.testing.assert_allclose(dest[:, :, q, I], calc[:, :, ng - 1, I], rtol=4e-16)
        np.testing.assert_allclose(dest[:, :, ng + ny + q, I], calc[:, :, ng + ny, I], rtol=4e-16)
    c = 0.5 * (dest[:, :, ng, 0] + dest[:, :, 0, ng])
    assert np.array_equal(dest[:, :, 0, 0], c)
    assert np.isfinite(dest).all()


def test_radiation_of_an_outgoing_wave_carries_the_boundary_value_out():
    """A field that changes only in x and moves towards -x with one cell per step (new(i) = old(i+1)): dTdt dTdx > 0 at the western
    side, dTde = 0, so the ghost value is (dTdx^2 old(-1) + Cx new(0)) / (dTdx^2 + Cx) with Cx = dTdt dTdx -- for a linear profile
    dTdt = -dTdx... checked here as the closed form, against the formula evaluated in numpy."""
    cfg, g = _cons_case()
    nx, ny, nz = cfg["ncell"]
    ng = 3
    x = np.arange(-ng, nx + ng, dtype=np.float64)
    prof = lambda s: np.sin(0.3 * s) + 0.1 * s
    calc = np.broadcast_to(prof(x), (3, nz, ny + 2 * ng, nx + 2 * ng)).copy()
    dest = np.broadcast_to(prof(x + 1.0), (3, nz, ny + 2 * ng, nx + 2 * ng)).copy()
    one = np.ones((1, 1, ny + 2 * ng + 1, nx + 2 * ng + 1))
    mf = O.omf(dest, (-ng, -ng, 0), (0, 0, 0), (ng, ng, 0))
    O.leaf("o_physbc_calc")(C.byref(g), C.byref(mf), O.ofab(one, (-ng, -ng, 0)), 0, 3, CONS, 0, O.ofab(calc, (-ng, -ng, 0)),
                            O.ofab(one, (-ng, -ng, 0)), O.ofab(one, (-ng, -ng, 0)))
    old = prof(x); new = prof(x + 1.0)
    i0 = ng                                                   # array position of dom_lo.x
    dTdt = old[i0] - new[i0]; dTdx = new[i0] - new[i0 + 1]
    if dTdt * dTdx < 0:
        dTdt = 0.0
    cff = max(dTdx * dTdx, 1e-20); Cx = dTdt * dTdx
    want = (cff * old[i0 - 1] + Cx * new[i0]) / (cff + Cx)
    assert np.array_equal(dest[:, :, ng:ng + ny, i0 - 1], np.full((3, nz, ny), want))
    assert Cx > 0.0                                           # the wave does leave through this side
