"""SolverChoice branches that no pinned case switches on: tracer_Hadv_scheme = centered4 (rh_params.tracer_c4), uv_Hadv_scheme =
centered2 (uv_c2), and use_prestep / use_uv3dmix / use_barotropic / use_salt = false.

Reference arms: REMORA_rhs_t_3d.cpp:128-141,165-177,225-240,260-275; REMORA_prestep_t_advection.cpp:222-234,255-266;
REMORA_rhs_uv_3d.cpp:148-157,302-311; REMORA_rhs_uv_2d.cpp:130,217; REMORA_Advance.cpp:13; REMORA_setup_step.cpp:214,315,346.

PARITY UNPINNED: every regression input of the reference keeps the defaults (upstream3 / upstream3, all switches on), so no gold
file covers these arms; the oracle's restatement of them stands on source reading.  What is checked here is that the HIP kernels
behind each flag (k_tracers_tiled<*, 1>, k_rhs_uv_tiled<1>, the uv_c2 arms of the barotropic kernels, the use_* guards of the
drivers) follow the oracle phase by phase, through the fused drivers and through the per-leaf entry points (RH_UNFUSED=1), and
that the flag really changed the run.
"""
import numpy as np
import pytest

from remora_amd import cases
from remora_amd.model import REMORA
from oracle.oracle_py import Oracle
from parity import upload_oracle_state, compare_states

pytestmark = pytest.mark.gpu

BITEXACT = {"Seamount", "DoubleGyre"}          # constant vertical mixing: no device transcendental on the path
RTOL_STEP = 1e-13                              # Advection: exp() of the analytic mixing profile, device libm vs glibc

FLAGS = [dict(tracer_c4=1), dict(uv_c2=1), dict(tracer_c4=1, uv_c2=1), dict(use_prestep=0), dict(use_uv3dmix=0),
         dict(use_barotropic=0), dict(use_salt=0)]


def _check(res, case, what):
    if case in BITEXACT:
        bad = {k: v for k, v in res.items() if v[0] != 0.0}
        assert not bad, f"{what}: not bit-identical: {bad}"
    else:
        bad = {k: v for k, v in res.items() if not (v[0] <= RTOL_STEP * max(v[1], 1e-300) or v[0] <= 1e-18)}
        assert not bad, f"{what}: beyond {RTOL_STEP}: {bad}"


def _run(case, flags, nsteps, what):
    cfg = cases.get(case)
    cfg.update(flags)
    orc = Oracle(cfg); rem = REMORA(cfg)
    if not cfg["use_prestep"]:
        # REMORA never initialises vec_sstore (make_new_level.cpp:493) and only prestep writes it: with use_prestep = false
        # rhs_t_3d reads whatever the allocation holds (signalling NaNs under the test inputs' fab.init_snan = 1; the oracle's
        # NaN fill reproduces that, and a comparison against NaN compares nothing).  Start both sides from the same defined
        # sstore instead, so that the guard of REMORA_setup_step.cpp:214 is checked on a defined result.
        ss = orc.field("sstore")[0]; cn = orc.field("cons_new")[0]
        ss[...] = cn[:, :, 1:-1, 1:-1]
    upload_oracle_state(orc, rem)
    assert rem.nfast == orc.nfast
    for step in range(nsteps):
        orc.begin_step(); rem.begin_step()
        orc.setup_step(); rem.setup_step()
        _check(compare_states(orc, rem), case, f"{what} step {step} setup_step")
        if cfg["use_barotropic"]:                                                   # REMORA_Advance.cpp:13
            for iif in range(orc.nfast + 1):
                orc.advance_2d_onestep(iif); rem.advance_2d_onestep(iif)
            _check(compare_states(orc, rem), case, f"{what} step {step} advance_2d")
        orc.advance_3d_ml(); rem.advance_3d_ml()
        res = compare_states(orc, rem)
        _check(res, case, f"{what} step {step} advance_3d_ml")
        assert all(np.isfinite(v[1]) for v in res.values())
        assert {"cons_new", "xvel_new", "yvel_new", "sstore", "Huon", "Hvom"} <= set(res), "a carried field had no defined value to compare"
        orc.end_step(); rem.end_step()
    out = {n: rem.get(n)[0].copy() for n in ("cons_new", "xvel_new", "yvel_new", "zeta")}
    rem.close(); orc.close()
    return out


def _ident(flags):
    return ",".join(f"{k}={v}" for k, v in flags.items())


@pytest.mark.parametrize("unfused", [0, 1])
@pytest.mark.parametrize("flags", FLAGS, ids=_ident)
@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre"])
def test_branch_phases_against_oracle(case, flags, unfused, monkeypatch):
    if unfused:
        monkeypatch.setenv("RH_UNFUSED", "1")
    _run(case, flags, 3, f"{case} {_ident(flags)} unfused={unfused}")


@pytest.mark.parametrize("flags", [dict(tracer_c4=1), dict(uv_c2=1)], ids=_ident)
def test_branch_flat_bathymetry(flags):
    """Advection runs remora.flat_bathymetry = true: rhs_t_3d's flat forms of both schemes (REMORA_rhs_t_3d.cpp:112-141,209-240),
    the predictor's centred flux that ignores the scheme (REMORA_prestep_t_advection.cpp:147-157)"""
    _run("Advection", flags, 3, f"Advection {_ident(flags)}")


@pytest.mark.parametrize("flags", [dict(tracer_c4=1), dict(uv_c2=1), dict(use_uv3dmix=0), dict(use_salt=0)], ids=_ident)
def test_branch_changes_the_run(flags):
    """the flag is not a no-op: three DoubleGyre steps with it differ from three steps without it (through rh_advance)"""
    outs = []
    for f in ({}, flags):
        cfg = cases.get("DoubleGyre"); cfg.update(f)
        rem = REMORA(cfg); rem.InitData(); rem.Evolve(3)
        outs.append({n: rem.get(n)[0].copy() for n in ("cons_new", "xvel_new", "yvel_new")})
        rem.close()
    assert any(not np.array_equal(outs[0][n], outs[1][n]) for n in outs[0]), f"{flags}: same bits as the default run"


def test_branches_through_rh_advance():
    """the single-call driver with both alternative schemes on, side-stream fills included"""
    cfg = cases.get("Seamount"); cfg.update(tracer_c4=1, uv_c2=1)
    orc = Oracle(cfg); rem = REMORA(cfg)
    upload_oracle_state(orc, rem)
    for step in range(3):
        orc.advance(1); rem.timeStep()
        _check(compare_states(orc, rem), "Seamount", f"c4+c2 rh_advance step {step}")
    rem.close(); orc.close()
