"""The A/B switches INTEGRATION.md advertises (read by rh_create) and the bench workload's physics at a many-tile size.

* RH_NOFUSE2D=1 (three-kernel barotropic call), RH_NOREGSOLVE=1 (LDS-ring column solver at every depth), RH_NO_ASYNC=1 (no side
  stream), RH_NO_BC2D=1 (ghost cells of the barotropic step through the batched BC passes instead of the rim blocks of k2d_fused) and RH_NO_MFFUSE=1 (update_massflux_3d as its own launches instead of on the velocity corrector's sweep) must give the same bits as the default path: each is checked against the oracle on the bit-exact cases.
* RH_VMIX_REWRITE=1: constant vertical mixing rewritten at every step and read from the arrays by the column solvers (the default
  writes Akv / Akt once per bound state and gives the solvers the constants in four L1-resident rows).
* RH_NO_CKONST=1: under constant mixing the register solvers read the constant rows like any array instead of taking the constant as
  a number (the default drops the coefficient loads and one division per level: FC(k) is CF(k-1) before its scaling, bit for bit).
* RH_NO_FLATSTRETCH=1: stretch_transform as the tiled kernel even when its arrays are dense over its box (the default then walks the
  plane linearly, 256 columns per block).
* RH_NO_XSHARE=1: k_rhs_uv_tiled evaluates all eight horizontal fluxes of a point itself (the default takes the two east ones from the
  neighbouring lane by a DPP shift and advances its tiles by 63 columns).
* RH_NO_K2MODE=1: k2d_fused always reads its time-stepping modes from its arguments (the default runs the steady predictor and corrector
  calls through instantiations that have them as constants).
* RH_NO_RHOFUSE=1: rho_eos (+ the hydrostatic pressure) and prsgrd as two launches with rho and P in memory (the default marches them as
  one kernel, k_rho_prsgrd, and keeps both in LDS).
* RH_POISON=1 fills the library's work arrays with NaN at the top of every step: a kernel that read scratch it had not written
  in the same step would produce NaN (and differ from the oracle).
* DoubleGyre physics, the bench workload's depth (50 levels) and boundary conditions (all walls), on 512 x 512 columns = 8 x 64
  tiles per side: one step against the oracle, every field incl. ghost cells, bit for bit.
* A seeded-random state (velocities, transports and tracers of both signs, ragged tile counts) pushed through the fused drivers.
"""
import numpy as np
import pytest

from remora_amd import cases
from remora_amd.model import REMORA
from oracle.oracle_py import Oracle
from parity import upload_oracle_state, compare_states

pytestmark = pytest.mark.gpu


def _bitexact(res, what):
    bad = {k: v for k, v in res.items() if v[0] != 0.0}
    assert not bad, f"{what}: not bit-identical: {bad}"


def _phases(cfg, nsteps, what):
    orc = Oracle(cfg); rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    for step in range(nsteps):
        orc.begin_step(); rem.begin_step()
        orc.setup_step(); rem.setup_step()
        _bitexact(compare_states(orc, rem), f"{what} step {step} setup_step")
        for iif in range(orc.nfast + 1):
            orc.advance_2d_onestep(iif); rem.advance_2d_onestep(iif)
        _bitexact(compare_states(orc, rem), f"{what} step {step} advance_2d")
        orc.advance_3d_ml(); rem.advance_3d_ml()
        _bitexact(compare_states(orc, rem), f"{what} step {step} advance_3d_ml")
        orc.end_step(); rem.end_step()
    rem.close(); orc.close()


@pytest.mark.parametrize("switch", ["RH_NOFUSE2D", "RH_NOREGSOLVE", "RH_POISON", "RH_NO_ASYNC", "RH_NO_BC2D", "RH_NO_MFFUSE", "RH_VMIX_REWRITE", "RH_NO_RHOFUSE", "RH_NO_CKONST", "RH_NO_FLATSTRETCH", "RH_NO_XSHARE", "RH_NO_K2MODE"])
@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre"])
def test_switch_keeps_the_bits(switch, case, monkeypatch):
    monkeypatch.setenv(switch, "1")
    _phases(cases.get(case), 3, f"{case} {switch}=1")


def test_poison_through_rh_advance(monkeypatch):
    """the single-call driver (rh_advance: side-stream fills included) with poisoned work arrays, three steps"""
    monkeypatch.setenv("RH_POISON", "1")
    cfg = cases.get("Seamount")
    orc = Oracle(cfg); rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    for step in range(3):
        orc.advance(1); rem.timeStep()
        _bitexact(compare_states(orc, rem), f"Seamount RH_POISON rh_advance step {step}")
    rem.close(); orc.close()


def test_doublegyre_bench_depth_many_tiles():
    cfg = cases.get("DoubleGyre")
    nx0, ny0, _ = cfg["ncell"]
    nx, ny, nz = 512, 512, 50
    # the bench's scaling (bench.py: bench_case): same basin, finer grid, dt by the CFL ratio
    cfg["ncell"] = (nx, ny, nz)
    cfg["fixed_dt"] = float(cfg["fixed_dt"] * min(nx0 / nx, ny0 / ny))
    _phases(cfg, 1, "DoubleGyre 512x512x50")


@pytest.mark.parametrize("case,ncell", [("Seamount", (150, 70, 13)), ("DoubleGyre", (140, 75, 6))])
def test_random_state_through_fused_drivers(case, ncell):
    cfg = cases.get(case)
    nx0, ny0, _ = cfg["ncell"]
    cfg["prob_hi"] = (cfg["prob_hi"][0] * ncell[0] / nx0, cfg["prob_hi"][1] * ncell[1] / ny0, cfg["prob_hi"][2])
    cfg["ncell"] = ncell
    orc = Oracle(cfg); rem = REMORA(cfg)
    rng = np.random.default_rng(12345)
    # perturb the oracle's own arrays in place (orc.field returns views), ghost cells included, then upload
    for name, amp in (("xvel_new", 0.3), ("yvel_new", 0.3), ("xvel_old", 0.3), ("yvel_old", 0.3), ("ubar", 0.05), ("vbar", 0.05),
                      ("zeta", 0.02), ("Zt_avg1", 0.02), ("ru", 1e-3), ("rv", 1e-3), ("ru2d", 1e-3), ("rv2d", 1e-3),
                      ("DU_avg1", 10.0), ("DV_avg1", 10.0), ("DU_avg2", 10.0), ("DV_avg2", 10.0), ("rubar", 1e-3), ("rvbar", 1e-3), ("rzeta", 1e-3)):
        a, _, _ = orc.field(name)
        m = np.isfinite(a)
        a[m] += rng.uniform(-amp, amp, a.shape)[m]
    for name in ("cons_new", "cons_old"):
        a, _, _ = orc.field(name)
        a[0][np.isfinite(a[0])] += rng.uniform(-1.0, 1.0, a[0].shape)[np.isfinite(a[0])]
        a[1][np.isfinite(a[1])] += rng.uniform(-0.1, 0.1, a[1].shape)[np.isfinite(a[1])]
        a[2][np.isfinite(a[2])] = rng.uniform(0.0, 1.0, a[2].shape)[np.isfinite(a[2])]
    upload_oracle_state(orc, rem)
    for step in range(2):
        orc.begin_step(); rem.begin_step()
        orc.setup_step(); rem.setup_step()
        _bitexact(compare_states(orc, rem), f"{case} random state step {step} setup_step")
        for iif in range(orc.nfast + 1):
            orc.advance_2d_onestep(iif); rem.advance_2d_onestep(iif)
        _bitexact(compare_states(orc, rem), f"{case} random state step {step} advance_2d")
        orc.advance_3d_ml(); rem.advance_3d_ml()
        res = compare_states(orc, rem)
        _bitexact(res, f"{case} random state step {step} advance_3d_ml")
        assert all(np.isfinite(v[1]) for v in res.values())
        orc.end_step(); rem.end_step()
    # both signs really occurred in the transports and the vertical velocity analogue
    hu, _, _ = rem.get("Huon"); hv, _, _ = rem.get("Hvom")
    assert (hu > 0).any() and (hu < 0).any() and (hv > 0).any() and (hv < 0).any()
    rem.close(); orc.close()


def test_rebind_moves_the_state_and_keeps_the_bits():
    """REMORA.rebind binds the state to another allocation under the same context: the same run from the new block must give the
    same bits, and going back to the first block as well (bench.py's placement search relies on it)."""
    import numpy as np
    from remora_amd import cases
    from remora_amd.model import REMORA
    rem = REMORA(cases.get("Seamount"))
    out = []
    first = None
    for leg in range(3):
        if leg == 1:
            first = rem.rebind()
            assert first.data_ptr() != rem.buf.data_ptr()
        if leg == 2:
            rem.rebind(first)
            assert rem.buf.data_ptr() == first.data_ptr()
        rem.InitData()
        rem.Evolve(3)
        out.append({n: rem.get(n)[0] for n in ("cons_new", "xvel_new", "yvel_new", "zeta", "ubar", "vbar", "Hz")})
    for n in out[0]:
        assert np.array_equal(out[0][n], out[1][n]) and np.array_equal(out[0][n], out[2][n]), n
    rem.close()


def test_const_vmix_arrays_hold_the_constants_and_survive_rebind():
    """constant vertical mixing (RH_VMIX_CONST) is written once per bound state: after steps from a fresh bind, after rebind to a
    zeroed block and after InitData the arrays must hold the constants (REMORA.cpp:565-573 rewrites them every step; a checkpoint
    or a plotfile reads them).  From the first step on the library takes the arrays to hold the constants: a caller who writes
    into them later must bind the state again (or run with RH_VMIX_REWRITE=1) -- include/remora_hip.h says so."""
    import numpy as np
    from remora_amd import cases
    from remora_amd.model import REMORA
    cfg = cases.get("Seamount")
    rem = REMORA(cfg)
    for leg in range(3):
        if leg == 1:
            rem.rebind()                                  # a zeroed block
        rem.InitData()
        rem.Evolve(2)
        akv = rem.get("Akv")[0]; akt = rem.get("Akt")[0]
        assert np.all(akv == 1.0e-5), leg
        assert np.all(akt[0] == 1.0e-6) and np.all(akt[1] == 1.0e-6) and np.all(akt[2] == 0.0), leg
    rem.close()
