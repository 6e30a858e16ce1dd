"""GPU parity tests proper: the HIP library (through its C ABI) against the CPU oracle on identical inputs, phase by
phase, and against the reference's own gold plotfiles after ten steps.

Tolerances.  The path is FP64 and built without FMA contraction, so every kernel reproduces the oracle's operation
order; the only non-identical arithmetic is exp() in the analytic vertical-mixing profile of the Upwelling/Advection
cases (device libm vs glibc, <= 1 ulp).  Cases without device transcendentals must therefore agree BIT FOR BIT with
the oracle; the others to 1e-13 of each field's maximum per step (north_star: ~1e-12 per step).  Gold plotfiles:
amrex_fcompare's criterion at the reference's own tolerance, rel 1e-11 / abs 1e-11 (Tests/CTestList.cmake:39).
"""
import numpy as np
import pytest

from remora_amd import cases
from remora_amd.model import REMORA
from oracle.oracle_py import Oracle
from parity import upload_oracle_state, compare_states, worst, COMPARE, GLS_FIELDS
from plotfile import read_plotfile, fcompare

pytestmark = pytest.mark.gpu

BITEXACT = {"Seamount", "DoubleGyre"}          # constant vertical mixing: no device transcendental on the path
RTOL_STEP = 1e-13


def _check(res, case, what):
    name, err, rel = worst(res)
    if case in BITEXACT:
        bad = {k: v for k, v in res.items() if v[0] != 0.0}
        assert not bad, f"{case} {what}: not bit-identical: {bad}"
    else:
        bad = {k: v for k, v in res.items() if not (v[0] <= RTOL_STEP * max(v[1], 1e-300) or v[0] <= 1e-18)}
        assert not bad, f"{case} {what}: beyond {RTOL_STEP}: {bad}"


@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre", "Upwelling", "Advection", "DoublyPeriodic", "DoublyPeriodic_bathy"])
def test_phases_against_oracle(case):
    """Three baroclinic steps from the oracle's initial state (covers the iic = 0, 1, >= 2 branches), comparing all
    fields incl. ghost cells after setup_step, after the barotropic loop and after advance_3d_ml."""
    cfg = cases.get(case)
    orc = Oracle(cfg)
    rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    assert rem.nfast == orc.nfast
    for step in range(3):
        orc.begin_step(); rem.begin_step()
        orc.setup_step(); rem.setup_step()
        _check(compare_states(orc, rem), case, f"step {step} setup_step")
        for iif in range(orc.nfast + 1):
            orc.advance_2d_onestep(iif); rem.advance_2d_onestep(iif)
        _check(compare_states(orc, rem), case, f"step {step} advance_2d")
        orc.advance_3d_ml(); rem.advance_3d_ml()
        _check(compare_states(orc, rem), case, f"step {step} advance_3d_ml")
        orc.end_step(); rem.end_step()
    rem.close(); orc.close()


@pytest.mark.parametrize("case,libm_pow", [("Upwelling_GLS", 0), ("Upwelling_GLS", 1), ("Channel_Test", 0)])
def test_gls_phases_against_oracle(case, libm_pow, monkeypatch):
    """Upwelling_GLS: Galperin stability functions, 16 levels; Channel_Test: Canuto_A, 50 levels, salinity-dependent density.
    libm_pow = 1: every per-point power through the device pow(); 0 (the default build): half-integer exponents through
    division / sqrt chains (rh_gls.hip: gpow).  Both must meet the same bar.
    vertical_mixing_type = gls (Upwelling_GLS): gls_prestep inside setup_step, gls_corrector inside advance_3d, compared with the
    oracle after every phase over four steps (iic = 0, 1, 2, 3: both time-level parities and the first-step branch), ghost cells
    included.  The closure calls pow() per point (device ocml vs glibc), so the bar is the north star's: 1e-12 of each field's
    maximum per step on the turbulence fields and what they feed (Akv, Akt -> u, v, tracers)."""
    monkeypatch.setenv("RH_GLS_LIBM_POW", str(libm_pow))
    cfg = cases.get(case)
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
            name, err, rel = worst(r)
            print(f"GLS step {step} {what}: worst {name} abs {err:.3e} rel {rel:.3e}")
            bad = {k: v for k, v in r.items() if not (v[0] <= 1e-12 * (step + 1) * max(v[1], 1e-300) or v[0] <= 1e-18)}
            assert not bad, f"{case} step {step} {what}: beyond 1e-12 per step: {bad}"
    rem.close(); orc.close()


@pytest.mark.parametrize("case", ["Seamount", "Advection", "Upwelling_GLS"])
def test_leaf_entry_points_against_oracle(case, monkeypatch):
    """RH_UNFUSED=1 makes the drivers go through the per-leaf C entry points (rh_prestep_t, rh_t3dmix, rh_rhs_t_3d, ...) instead
    of the fused multi-tracer kernels: both routes must agree with the oracle (and hence with each other)."""
    monkeypatch.setenv("RH_UNFUSED", "1")
    cfg = cases.get(case)
    orc = Oracle(cfg); rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    for step in range(3):
        orc.advance(1); rem.timeStep()
        _check(compare_states(orc, rem, COMPARE + (GLS_FIELDS if "gls" in cfg else [])), case, f"unfused step {step}")
    rem.close(); orc.close()


@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre"])
def test_mask_reading_path(case):
    """masks_all_one = 0 makes every kernel read mskr/msku/mskv/mskp and run the final cons *= mskr pass
    (REMORA_advance_3d_ml.cpp:51-62); with the all-ones masks of the pinned cases the result must not change by a bit."""
    cfg = cases.get(case)
    orc = Oracle(cfg)
    cfg_m = dict(cfg, masks_all_one=0)
    rem = REMORA(cfg_m)
    upload_oracle_state(orc, rem)
    for step in range(2):
        orc.advance(1); rem.timeStep()
        _check(compare_states(orc, rem), case, f"masked step {step}")
    rem.close(); orc.close()


@pytest.mark.parametrize("case", ["Upwelling", "Seamount", "DoubleGyre", "Advection", "DoublyPeriodic", "DoublyPeriodic_bathy", "Upwelling_GLS", "Channel_Test"])
def test_gold_plotfile(case):
    """Device-side initial state + ten steps through REMORA::Advance == the reference's regression test."""
    names, gold = read_plotfile(case)
    rem = REMORA(cases.get(case))
    rem.InitData()
    rem.Evolve(10)
    mine = rem.plot(with_scalar=(len(names) == 6))
    res = fcompare(gold, mine)
    assert all(r[2] for r in res), f"{case}: {list(zip(names, res))}"
    rem.close()


@pytest.mark.parametrize("case,from_oracle_init,bound", [("Seamount", True, 1e-12), ("Upwelling", True, 1e-11),
                                                         ("Seamount", False, 1e-9), ("Upwelling", False, 1e-9)])
def test_drift_vs_oracle(case, from_oracle_init, bound):
    """Drift of the plotfile variables over 20 steps of the full path against the oracle, reported per checkpoint.
    from_oracle_init: both start from the oracle's initial state, so only the hot path differs (Upwelling: exp() in the
    vertical-mixing profile).  Otherwise the device builds its own initial state (tanh/exp of the bathymetry and the
    stratification differ from glibc by <= 1 ulp and the difference is carried along)."""
    cfg = cases.get(case)
    orc = Oracle(cfg); rem = REMORA(cfg)
    if from_oracle_init:
        upload_oracle_state(orc, rem)
    else:
        rem.InitData()
    worst_rel = 0.0
    for step in range(20):
        orc.advance(1); rem.timeStep()
        if step in (0, 9, 19):
            a = orc.plot(); b = rem.plot()
            rel = []
            for c in range(a.shape[0]):
                mx = np.max(np.abs(a[c]))
                rel.append(float(np.max(np.abs(a[c] - b[c])) / mx) if mx > 0 else 0.0)
            print(f"{case} from_oracle_init={from_oracle_init} step {step + 1}: rel drift per plot var = {['%.2e' % r for r in rel]}")
            worst_rel = max(worst_rel, max(rel))
    assert worst_rel < bound
    rem.close(); orc.close()


def test_rccl_transport_selftest():
    """The built-in RCCL transport on a one-rank communicator: unique id, ncclCommInitRank, grouped send/recv to self, and
    the message order a two-rank periodic exchange relies on (more ranks than GPUs cannot be formed on a one-GPU box)."""
    import ctypes as C
    from remora_amd import lib as L
    rem = REMORA(cases.get("Seamount"))
    idbuf = (C.c_ubyte * 128)()
    L.check(rem.lib.rh_comm_unique_id(idbuf))
    L.check(rem.lib.rh_comm_init(rem.ctx, idbuf, 0, 1))
    L.check(rem.lib.rh_comm_selftest(rem.ctx, 100000))
    rem.close()


@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre", "Advection"])
def test_reductions_against_oracle(case):
    """estTimeStep (REMORA_ComputeTimestep.cpp:39-95) and sum_integrated_quantities (REMORA_SumIQ.cpp:9-141) after five steps from
    the same state.  Maxima (the cfl dt, the maximum speed) are order-independent: bit-identical.  The three volume-weighted sums
    are folded in another order than the reference's serial MultiFab::Dot: relative 1e-13 (about n * eps / sqrt(n) of the
    2e4..1e5 cells of these cases, with margin)."""
    cfg = cases.get(case)
    orc = Oracle(cfg)
    rem = REMORA(cfg, device="cuda:0")
    upload_oracle_state(orc, rem)
    for _ in range(5):
        orc.advance(1); rem.timeStep()
    if case in BITEXACT:
        assert rem.estTimeStep(0.8, local_only=True) == orc.est_timestep(0.8)
    else:
        assert abs(rem.estTimeStep(0.8, local_only=True) / orc.est_timestep(0.8) - 1.0) < 1e-12
    assert rem.estTimeStep(0.8) == cfg["fixed_dt"]                    # fixed_dt wins (:90-91)
    a = rem.sum_integrated_quantities(); b = orc.sum_iq()
    rtol = 1e-13 if case in BITEXACT else 1e-11          # Advection: the fields themselves differ by the device exp() (see the module docstring)
    for x, y, name in zip(a[:3], b[:3], ("scalar", "kineng", "volume")):
        assert abs(x - y) <= rtol * max(abs(y), 1e-300) or abs(x - y) < 1e-300, (name, x, y)
    if case in BITEXACT:
        assert a[3] == b[3]
    else:
        assert abs(a[3] - b[3]) <= 1e-12 * abs(b[3])
    rem.close(); orc.close()


@pytest.mark.parametrize("nz", [50, 64, 20, 70])
def test_deep_columns_against_oracle(nz):
    """The column kernels pick their form by depth: register-resident solver for the depths it is built for (13, 16, 50 and 64, the depth of BASELINE
    configs[4]), LDS ring kernel otherwise (20), with the dynamic-LDS opt-in above 62 levels (70).  Seamount physics
    (steep bathymetry, no device transcendental: bit-exact) on its own horizontal grid with nz levels, three steps, all fields."""
    cfg = cases.get("Seamount")
    cfg["ncell"] = (cfg["ncell"][0], cfg["ncell"][1], nz)
    orc = Oracle(cfg); rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    for step in range(3):
        orc.advance(1); rem.timeStep()
        _check(compare_states(orc, rem), "Seamount", f"nz {nz} step {step}")
    rem.close(); orc.close()


@pytest.mark.parametrize("case,ncell", [("Upwelling_GLS", (150, 133, 12)), ("Seamount", (197, 135, 13)), ("DoubleGyre", (140, 150, 6))])
def test_many_tiles_against_oracle(case, ncell):
    """Every gold case is at most 81 cells wide, i.e. one or two 64-lane tiles in i.  The same physics on grids of several tiles
    in both directions with ragged last tiles (150 = 2 x 64 + 22, 197 = 3 x 63 + 8 for the overlapped-lane kernels), three steps,
    all fields incl. ghost cells: exercises the tile seams, the XCD-aware tile order and the patch order of the k-marching kernels."""
    cfg = cases.get(case)
    nx0, ny0, _ = cfg["ncell"]
    cfg["prob_hi"] = (cfg["prob_hi"][0] * ncell[0] / nx0, cfg["prob_hi"][1] * ncell[1] / ny0, cfg["prob_hi"][2])   # same dx, dy
    cfg["ncell"] = ncell
    orc = Oracle(cfg); rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    fields = COMPARE + (GLS_FIELDS if "gls" in cfg else [])
    for step in range(3):
        orc.advance(1); rem.timeStep()
        res = compare_states(orc, rem, fields)
        if "gls" in cfg:
            bad = {k: v for k, v in res.items() if not (v[0] <= 1e-12 * (step + 1) * max(v[1], 1e-300) or v[0] <= 1e-18)}
            assert not bad, f"{case} {ncell} step {step}: {bad}"
        else:
            _check(res, case, f"{ncell} step {step}")
    rem.close(); orc.close()
