"""Land / sea masks on the hot path: the HIP library against the oracle with land in the domain.

The reference reads mask_rho / mask_u / mask_v from its NetCDF grid file (Initialization/REMORA_init_from_netcdf.cpp:372-412), derives
mskp from mskr (REMORA::update_mskp, make_new_level.cpp:617-655) and multiplies by the masks inside nearly every pass of the step.  Its
regression cases all have masks of one, so PARITY WITH THE REFERENCE IS UNPINNED for this path: the oracle restates the reference's
mask factors line by line, and these tests hold the device to the oracle (bit for bit where the case has no device transcendental).
The land laid out here has an island, a one-cell island, two cells touching across a diagonal (mskp = 0 there), land on a wall /
across a periodic side, and a straight coast (mskp = 2)."""
import numpy as np
import pytest

from remora_amd import cases
from remora_amd.model import REMORA
from oracle.oracle_py import Oracle
from parity import upload_oracle_state, compare_states, worst, COMPARE, GLS_FIELDS
from landmask import water, mask_arrays

pytestmark = pytest.mark.gpu

BITEXACT = {"Seamount", "DoubleGyre"}


def put_masks(orc, rem, cfg):
    """mskr, msku = mskr(i-1) * mskr(i), mskv = mskr(j-1) * mskr(j) on both sides; mskp by each side's own update_mskp."""
    f = water(cfg)
    for name, di, dj in (("mskr", 0, 0), ("msku", 1, 0), ("mskv", 0, 1)):
        a, lo, hi = orc.field(name)
        m = mask_arrays(f, lo, hi, di, dj)
        a[0, 0] = m
        rem.set(name, m[None, None], lo)
    rem.set("mskp", np.ones_like(rem.get("mskp")[0]))          # init_masks (make_new_level.cpp:381-389); the oracle's start at one too
    orc.update_mskp(); rem.update_mskp()
    for name in ("mskr", "msku", "mskv", "mskp"):
        a, alo, ahi = orc.field(name)
        d, dlo, dhi = rem.get(name)
        assert alo == tuple(dlo) and ahi == tuple(dhi), name
        assert np.array_equal(a, d), f"{name}: {np.argwhere(a != d)[:5]}"
    p = orc.field("mskp")[0]
    assert (p == 0.0).any() and (p == 2.0).any() and (p == 1.0).any()
    # a start at rest on land: velocities carry their masks (what a masked grid file would hold)
    for name, mk in (("xvel_old", "msku"), ("xvel_new", "msku"), ("yvel_old", "mskv"), ("yvel_new", "mskv"), ("ubar", "msku"), ("vbar", "mskv")):
        a, alo, ahi = orc.field(name)
        m, mlo, mhi = orc.field(mk)
        sub = m[0, 0, alo[1] - mlo[1]:ahi[1] - mlo[1] + 1, alo[0] - mlo[0]:ahi[0] - mlo[0] + 1]
        fin = np.isfinite(a)
        a[fin] = (a * sub[None, None])[fin]


def _check(res, case, what, step):
    if case in BITEXACT:
        bad = {k: v for k, v in res.items() if v[0] != 0.0}
        assert not bad, f"{case} {what}: not bit-identical: {bad}"
    else:
        bad = {k: v for k, v in res.items() if not (v[0] <= 1e-12 * (step + 1) * max(v[1], 1e-300) or v[0] <= 1e-18)}
        assert not bad, f"{case} {what}: beyond 1e-12 per step: {bad}"


@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre", "Upwelling", "Upwelling_GLS", "Channel_Test_orlanski"])
def test_land_masks_phases_against_oracle(case):
    """Four steps with land in the domain, every field compared (ghost cells included) after setup_step, after the barotropic loop
    and after advance_3d_ml."""
    cfg = dict(cases.get(case), masks_all_one=0)
    orc = Oracle(cfg); rem = REMORA(cfg)
    put_masks(orc, rem, cfg)
    upload_oracle_state(orc, rem)
    fields = COMPARE + (GLS_FIELDS if "gls" in cfg else [])
    for step in range(4):
        orc.begin_step(); rem.begin_step()
        orc.setup_step(); rem.setup_step()
        _check(compare_states(orc, rem, fields), case, f"step {step} setup_step", step)
        for iif in range(orc.nfast + 1):
            orc.advance_2d_onestep(iif); rem.advance_2d_onestep(iif)
        _check(compare_states(orc, rem, fields), case, f"step {step} advance_2d", step)
        orc.advance_3d_ml(); rem.advance_3d_ml()
        res = compare_states(orc, rem, fields)
        _check(res, case, f"step {step} advance_3d_ml", step)
        orc.end_step(); rem.end_step()
    # the land stays at rest and the state stays finite
    u, ulo, _ = rem.get("xvel_new"); mu, mlo, _ = rem.get("msku")
    sub = mu[0, 0, ulo[1] - mlo[1]:ulo[1] - mlo[1] + u.shape[2], ulo[0] - mlo[0]:ulo[0] - mlo[0] + u.shape[3]]
    nb = cfg["ncell"]
    inner = (slice(None), slice(None), slice(2, 2 + nb[1]), slice(2, 2 + nb[0] + 1))
    assert np.all(u[inner][:, :, sub[2:2 + nb[1], 2:2 + nb[0] + 1] == 0.0] == 0.0)
    assert np.isfinite(rem.get("cons_new")[0]).all()
    rem.close(); orc.close()


@pytest.mark.parametrize("case", ["Seamount", "DoubleGyre"])
@pytest.mark.parametrize("switch", ["RH_UNFUSED", "RH_NOFUSE2D", "RH_NOREGSOLVE", "RH_NO_MFFUSE", "RH_NO_BC2D"])
def test_land_masks_fused_and_leaf_paths_agree(case, switch, monkeypatch):
    """The same run with one of the fusion switches off must give the same bits: the masks enter the fused kernels and the
    per-leaf kernels at the same places."""
    cfg = dict(cases.get(case), masks_all_one=0)
    out = []
    for off in (False, True):
        if off:
            monkeypatch.setenv(switch, "1")
        orc = Oracle(cfg); rem = REMORA(cfg)
        put_masks(orc, rem, cfg)
        upload_oracle_state(orc, rem)
        rem.Evolve(3)
        out.append({n: rem.get(n)[0] for n in COMPARE})
        rem.close(); orc.close()
    for n in COMPARE:
        assert np.array_equal(out[0][n], out[1][n], equal_nan=True), n
