"""Open boundaries: REMORA::init_bcs for every kind bc.<>.type accepts and the physical-BC routines
(impose_cons_bcs / impose_xvel_bcs / impose_yvel_bcs) for every mathematical type, HIP (through the C ABI) against the CPU oracle,
BIT FOR BIT (copies, sign flips and the radiation formula's + - * / in the reference's order; no transcendental).

PARITY UNPINNED: the reference holds no gold file or test for a run with open sides (Exec/Channel_Test/inputs_orlanski is not in
Tests/CTestList.cmake), so oracle/bc.c's restatement of these branches is checked against the HIP path and against hand-derived
expectations (tests/test_open_bc_oracle.py), not against output of the reference.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from remora_amd import cases, lib as L
from remora_amd.cases import bc_per_variable, bc_per_side, XLO, YLO, XHI, YHI
from remora_amd.model import REMORA
from oracle import oracle_py as O
from oracle.oracle_py import Oracle
from parity import upload_oracle_state, compare_states, COMPARE, GLS_FIELDS
from test_gpu_leaves import Arr, _a4

pytestmark = pytest.mark.gpu

NX, NY, NZ = 13, 11, 4
BCV = dict(CONS=0, XVEL=3, YVEL=4, ZVEL=5, UBAR=6, VBAR=7, ZETA=8, TKE=9, FOEXTRAP_PERIODIC=10, FOEXTRAP=11, U2D=12, V2D=13)

_orl, _out, _cl, _sym, _nud = ("orlanski_rad",) * 4, ("outflow",) * 4, ("clamped",) * 4, ("symmetry",) * 4, ("orlanski_rad_nudg",) * 4
CONFIGS = {
    # Exec/Channel_Test/inputs_orlanski
    "orlanski": ((0, 0), bc_per_variable(temp=_orl, salt=_orl, scalar=_orl, u=_orl, v=_orl, ubar=_out, vbar=_out, zeta=_out, tke=_out)),
    "clamped": ((0, 0), bc_per_variable(temp=_cl, salt=_cl, scalar=_cl, u=_cl, v=_cl, ubar=_cl, vbar=_cl, zeta=_cl, tke=_out)),
    "symmetry": ((0, 0), bc_per_side("symmetry", "symmetry", "symmetry", "symmetry")),
    "walls_and_inflow": ((0, 0), bc_per_side(xlo="inflow", xhi="outflow", ylo="noslipwall", yhi="slipwall")),
    "flather": ((0, 0), bc_per_variable(temp=_nud, salt=_nud, scalar=_nud, u=_nud, v=_nud, ubar=("flather",) * 4, vbar=("flather",) * 4,
                                        zeta=("chapman",) * 4, tke=_out)),
    # a different kind on every face (words go to xlo, yhi, xhi, ylo)
    "mixed": ((0, 0), bc_per_variable(temp=("orlanski_rad", "clamped", "outflow", "symmetry"), salt=("clamped", "orlanski_rad", "symmetry", "outflow"),
                                      scalar=("inflow", "orlanski_rad", "orlanski_rad_nudg", "orlanski_rad"),
                                      u=("orlanski_rad", "outflow", "clamped", "symmetry"), v=("symmetry", "orlanski_rad", "orlanski_rad", "clamped"),
                                      ubar=("clamped", "outflow", "flather", "slipwall"), vbar=("outflow", "flather", "inflow", "clamped"),
                                      zeta=("chapman", "clamped", "outflow", "symmetry"), tke=_out)),
    # the channel of the reference's own Channel_Test, with its two walls opened
    "x_periodic": ((1, 0), bc_per_variable(temp=("periodic", "orlanski_rad", "periodic", "orlanski_rad"), salt=("periodic", "clamped", "periodic", "orlanski_rad"),
                                           scalar=("periodic", "orlanski_rad", "periodic", "outflow"), u=("periodic", "orlanski_rad", "periodic", "orlanski_rad"),
                                           v=("periodic", "orlanski_rad", "periodic", "orlanski_rad"), w=("periodic", "outflow", "periodic", "outflow"),
                                           ubar=("periodic", "outflow", "periodic", "outflow"), vbar=("periodic", "outflow", "periodic", "outflow"),
                                           zeta=("periodic", "outflow", "periodic", "outflow"), tke=("periodic", "outflow", "periodic", "outflow"))),
}


def _cfg(name):
    per, bc = CONFIGS[name]
    cfg = cases.get("DoubleGyre")
    cfg["ncell"] = (NX, NY, NZ)
    cfg["periodic"] = per
    cfg["bc"] = bc
    if name == "walls_and_inflow":
        cfg["bc_val"] = {"u": {XLO: 0.3, YLO: 0.2}, "v": {XLO: -0.1, YLO: 0.7}, "scalar": {XLO: 2.5}}
    return cfg


@pytest.mark.parametrize("name", list(CONFIGS))
def test_bc_table_matches_oracle(name):
    """REMORA::init_bcs: domain_bcs_type of all fourteen BCVars entries"""
    cfg = _cfg(name)
    rem = REMORA(cfg)
    g = O.geom(cfg)
    for comp in range(14):
        out = (C.c_int * 6)()
        L.check(rem.lib.rh_get_bc(rem.ctx, comp, out))
        assert list(out) == O.get_bc(g, comp), (name, comp)
    rem.close()


def test_set_bc_rejects_what_init_bcs_aborts_on():
    cfg = cases.get("DoubleGyre"); cfg["ncell"] = (NX, NY, NZ)
    rem = REMORA(cfg)
    lib = rem.lib
    P = cases.PBC
    assert lib.rh_set_bc(rem.ctx, BCV["UBAR"], XLO, P["orlanski_rad"], 0.0) != 0          # :539
    assert lib.rh_set_bc(rem.ctx, BCV["ZETA"], XLO, P["orlanski_rad"], 0.0) != 0          # :624
    assert lib.rh_set_bc(rem.ctx, BCV["CONS"], XLO, P["chapman"], 0.0) != 0               # :71-73
    assert lib.rh_set_bc(rem.ctx, BCV["XVEL"], XLO, P["flather"], 0.0) != 0               # :80-82
    assert lib.rh_set_bc(rem.ctx, BCV["CONS"], XLO, P["periodic"], 0.0) != 0              # :96-98
    assert lib.rh_set_bc(rem.ctx, BCV["CONS"], 2, P["outflow"], 0.0) != 0                 # z faces are not settable
    # a refused call leaves the table as it was
    out = (C.c_int * 6)(); L.check(lib.rh_get_bc(rem.ctx, BCV["UBAR"], out))
    assert list(out) == [3, 2, 2, 3, 2, 2]
    rem.close()
    cfg = cases.get("Seamount"); cfg["ncell"] = (NX, NY, NZ)
    rem = REMORA(cfg)
    assert rem.lib.rh_set_bc(rem.ctx, BCV["CONS"], XLO, P["outflow"], 0.0) != 0           # :107-115
    rem.close()


class Field:
    """A random array on the grown box of a given index type, its calc twin, and both sides' views of them"""

    def __init__(self, rng, ix, nc, ng, nk, calc_ng=None):
        self.ix, self.ng = ix, (ng, ng, 0)
        lo = (-ng, -ng, 0)
        shape = (nc, nk, NY + ix[1] + 2 * ng, NX + ix[0] + 2 * ng)
        self.dest = Arr(rng.uniform(-1.0, 1.0, shape), lo)
        cg = ng if calc_ng is None else calc_ng
        cshape = (nc, nk, NY + ix[1] + 2 * cg, NX + ix[0] + 2 * cg)
        self.calc = Arr(rng.uniform(-1.0, 1.0, cshape), (-cg, -cg, 0))

    def mf(self):
        return O.omf(self.dest.h, self.dest.lo, self.ix, self.ng)


def _mask(rng, ix):
    g = 3
    a = (rng.uniform(0, 1, (1, 1, NY + ix[1] + 2 * g, NX + ix[0] + 2 * g)) > 0.2).astype(np.float64)
    return Arr(a, (-g, -g, 0))


def _run(rem, g, f, msk, icomp, ncomp, bccomp, nnf, calc, msku, mskv, what):
    ng3 = (C.c_int * 3)(*f.ng)
    null4 = L.Array4()
    O.leaf("o_physbc_calc")(C.byref(g), C.byref(f.mf()), msk.fab(), icomp, ncomp, bccomp, nnf,
                            calc.fab() if calc is not None else O.null_fab(), msku.fab() if msku else O.null_fab(), mskv.fab() if mskv else O.null_fab())
    L.check(rem.lib.rh_physbc_calc(rem.ctx, f.dest.a4(), ng3, msk.a4(), icomp, ncomp, bccomp, nnf,
                                   calc.a4() if calc is not None else null4, msku.a4() if msku else null4, mskv.a4() if mskv else null4))
    torch.cuda.synchronize()
    assert f.dest.same(), what
    assert np.isfinite(f.dest.h).all(), what


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("inplace", [False, True])
def test_physbc_every_kind(name, inplace):
    """REMORAPhysBCFunct::operator(): cell-centred 3-D (cons, and sstore's ghost width 2), x faces, y faces, and the 2-D barotropic
    arrays (zeta, ubar, vbar: one component of three), with random land masks, each against the oracle.  inplace: calc is the filled
    array itself (REMORA::InitData)."""
    cfg = _cfg(name)
    rem = REMORA(cfg)
    g = O.geom(cfg)
    rng = np.random.default_rng(2024 + len(name))
    mskr, msku, mskv = _mask(rng, (0, 0)), _mask(rng, (1, 0)), _mask(rng, (0, 1))
    jobs = [("cons", Field(rng, (0, 0, 0), 3, 3, NZ), mskr, 0, 3, BCV["CONS"]),
            ("sstore", Field(rng, (0, 0, 0), 3, 2, NZ, calc_ng=3), mskr, 0, 3, BCV["CONS"]),
            ("salt only", Field(rng, (0, 0, 0), 3, 3, NZ), mskr, 1, 1, BCV["CONS"] + 1),
            ("xvel", Field(rng, (1, 0, 0), 1, 3, NZ), msku, 0, 1, BCV["XVEL"]),
            ("yvel", Field(rng, (0, 1, 0), 1, 3, NZ), mskv, 0, 1, BCV["YVEL"]),
            ("zeta", Field(rng, (0, 0, 0), 3, 3, 1), mskr, 1, 1, BCV["ZETA"]),
            ("ubar", Field(rng, (1, 0, 0), 3, 2, 1), msku, 2, 1, BCV["UBAR"]),
            ("vbar", Field(rng, (0, 1, 0), 3, 2, 1), mskv, 0, 1, BCV["VBAR"]),
            ("bustr", Field(rng, (1, 0, 0), 1, 2, 1), msku, 0, 1, BCV["U2D"]),
            ("bvstr", Field(rng, (0, 1, 0), 1, 2, 1), mskv, 0, 1, BCV["V2D"])]
    for what, f, msk, icomp, ncomp, bccomp in jobs:
        if inplace and what == "sstore":
            continue
        calc = f.dest if inplace else f.calc
        before = f.dest.h.copy()
        _run(rem, g, f, msk, icomp, ncomp, bccomp, 0, calc, msku, mskv, (name, what, inplace))
        if what in ("cons", "xvel", "yvel"):
            assert not np.array_equal(before, f.dest.h), (name, what, "nothing was filled")
        # the valid region is never touched, except the boundary faces of a face-centred array
        v = f.dest.h[:, :, f.ng[1] + 1:-f.ng[1] - 1, f.ng[0] + 1:-f.ng[0] - 1]
        assert np.array_equal(v, before[:, :, f.ng[1] + 1:-f.ng[1] - 1, f.ng[0] + 1:-f.ng[0] - 1]), (name, what)
    rem.close()


def test_orlanski_without_calc_is_an_error():
    """the reference reads an empty Array4 there (undefined behaviour); this library refuses"""
    cfg = _cfg("orlanski")
    rem = REMORA(cfg)
    rng = np.random.default_rng(5)
    f = Field(rng, (0, 0, 0), 3, 3, NZ); mskr = _mask(rng, (0, 0))
    ng3 = (C.c_int * 3)(*f.ng)
    assert rem.lib.rh_physbc(rem.ctx, f.dest.a4(), ng3, mskr.a4(), 0, 3, BCV["CONS"]) != 0
    assert b"calc" in rem.lib.rh_last_error()
    rem.close()


def test_n_not_fill_with_open_sides():
    """n_not_fill = 1 (the ghost line next to the domain is left alone) with copy-like and reflecting sides"""
    for name in ("clamped", "symmetry", "walls_and_inflow"):
        cfg = _cfg(name)
        rem = REMORA(cfg); g = O.geom(cfg)
        rng = np.random.default_rng(77)
        f = Field(rng, (0, 0, 0), 2, 3, 1); mskr = _mask(rng, (0, 0))
        _run(rem, g, f, mskr, 0, 2, BCV["CONS"], 1, None, None, None, (name, "n_not_fill"))
        rem.close()


def test_init_state_with_open_sides():
    """rh_init_state = REMORA::InitData: the three FillPatch calls of REMORA.cpp:293-295 run the radiation passes IN PLACE.
    Tracers, v and w must come out bit-identical; the channel's initial u goes through a device transcendental (prob.cpp's profile,
    1 ulp from glibc with closed walls as well), so u is held to one ulp of its maximum here and the in-place ordering of the u
    passes is what test_physbc_every_kind[inplace] pins bit for bit."""
    cfg = cases.get("Channel_Test_orlanski")
    orc = Oracle(cfg); rem = REMORA(cfg)
    rem.InitData()
    res = compare_states(orc, rem, ["cons_new", "cons_old", "xvel_new", "yvel_new", "zvel_new", "xvel_old", "yvel_old"])
    bad = {k: v for k, v in res.items() if v[0] > (2.3e-16 * v[1] if k.startswith("xvel") else 0.0)}
    assert not bad, bad
    rem.close(); orc.close()


def test_channel_orlanski_phases_against_oracle():
    """Exec/Channel_Test/inputs_orlanski: four open sides, tracers and 3-D momentum radiating, GLS closure.  Four steps, every field
    with its ghost cells after setup_step and after advance_3d_ml.  Bar: Channel_Test's (the closure calls pow() per point: 1e-12 of
    each field's maximum per step, test_gpu_parity.py); the boundary passes themselves are bit-exact (test_physbc_every_kind)."""
    cfg = cases.get("Channel_Test_orlanski")
    orc = Oracle(cfg); rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    fields = COMPARE + GLS_FIELDS
    for step in range(4):
        orc.begin_step(); rem.begin_step()
        orc.setup_step(); rem.setup_step()
        res = compare_states(orc, rem, fields)
        for iif in range(orc.nfast + 1):
            orc.advance_2d_onestep(iif); rem.advance_2d_onestep(iif)
        orc.advance_3d_ml(); rem.advance_3d_ml()
        res2 = compare_states(orc, rem, fields)
        orc.end_step(); rem.end_step()
        for what, r in (("setup_step", res), ("advance_3d_ml", res2)):
            bad = {k: v for k, v in r.items() if not (v[0] <= 1e-12 * (step + 1) * max(v[1], 1e-300) or v[0] <= 1e-18)}
            assert not bad, f"step {step} {what}: {bad}"
    rem.close(); orc.close()


def test_doublegyre_open_east_bit_exact():
    """No transcendental on the path (constant mixing): DoubleGyre with its eastern and northern walls replaced by radiating /
    zero-gradient sides must follow the oracle BIT FOR BIT through three whole steps (k2d_fused's wall branch must switch itself off)."""
    cfg = cases.get("DoubleGyre")
    cfg["bc"] = bc_per_variable(temp=("slipwall", "orlanski_rad", "orlanski_rad", "slipwall"), salt=("slipwall", "orlanski_rad", "orlanski_rad", "slipwall"),
                                scalar=("slipwall", "clamped", "orlanski_rad", "slipwall"), u=("slipwall", "orlanski_rad", "orlanski_rad", "slipwall"),
                                v=("slipwall", "orlanski_rad", "orlanski_rad", "slipwall"), ubar=("slipwall", "outflow", "outflow", "slipwall"),
                                vbar=("slipwall", "outflow", "outflow", "slipwall"), zeta=("slipwall", "outflow", "outflow", "slipwall"),
                                tke=("slipwall", "outflow", "outflow", "slipwall"))
    orc = Oracle(cfg); rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    for step in range(3):
        orc.advance(); rem.timeStep()
        res = compare_states(orc, rem)
        bad = {k: v for k, v in res.items() if v[0] != 0.0}
        assert not bad, f"step {step}: {bad}"
    rem.close(); orc.close()
