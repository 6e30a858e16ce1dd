"""Error behaviour of the C ABI on the GPU box: the reference's only error channel is amrex::Abort; here
every entry point returns non-zero with a message in rh_last_error() BEFORE anything is launched, so a wrong shape or mode is a
Python exception in these tests and never a GPU memory fault."""
import ctypes as C

import pytest
import torch

from remora_amd import cases, lib as L
from remora_amd.model import REMORA, make_params

pytestmark = pytest.mark.gpu


def _a4(t, lo):
    a = L.Array4()
    a.p = t.data_ptr()
    nc, nk, nj, ni = t.shape
    for d, (l, n) in enumerate(zip(lo, (ni, nj, nk))):
        a.lo[d] = l; a.hi[d] = l + n - 1
    a.ncomp = nc
    return a


def test_operand_smaller_than_the_box_is_rejected():
    """rh_vert_visc_3d with an Akv array that lacks the k = N+1 face, and rh_omega with a W array without the grow-1 rim."""
    rem = REMORA(cases.get("Seamount"))
    nx, ny, nz = rem.ncell
    phi = torch.zeros((1, nz, ny + 4, nx + 4), dtype=torch.float64, device="cuda:0")
    Hz = torch.ones_like(phi)
    Akv_short = torch.zeros((1, nz, ny + 4, nx + 4), dtype=torch.float64, device="cuda:0")       # needs nz + 1 faces
    bx = L.box((0, 0, 0), (nx - 1, ny - 1, nz - 1))
    with pytest.raises(L.RemoraHipError, match="does not cover box"):
        L.check(rem.lib.rh_vert_visc_3d(rem.ctx, bx, 0, 0, _a4(phi, (-2, -2, 0)), _a4(Hz, (-2, -2, 0)), _a4(Akv_short, (-2, -2, 0)), nz - 1, 60.0))
    W_small = torch.zeros((1, nz + 1, ny, nx), dtype=torch.float64, device="cuda:0")
    g1 = L.box((-1, -1, 0), (nx, ny, nz - 1))
    s = rem.state
    with pytest.raises(L.RemoraHipError, match="does not cover box"):
        L.check(rem.lib.rh_omega(rem.ctx, g1, _a4(W_small, (0, 0, 0)), s.Huon, s.Hvom, s.z_w, s.h, nz - 1))
    rem.sync()                                   # nothing was launched, the context is still usable
    rem.InitData(); rem.timeStep()
    rem.close()


def test_array_too_large_for_32_bit_indexing_is_rejected():
    """Kernels index with 32 bits: a view of 2^32 elements or more must be refused by the guard (no memory behind it is touched)."""
    rem = REMORA(cases.get("Seamount"))
    nx, ny, nz = rem.ncell
    small = torch.zeros((1, nz + 1, ny + 4, nx + 4), dtype=torch.float64, device="cuda:0")
    huge = _a4(small, (-2, -2, 0))
    huge.hi[0] = huge.lo[0] + 70000 - 1; huge.hi[1] = huge.lo[1] + 70000 - 1      # 4.9e9 "elements" per level
    bx = L.box((0, 0, 0), (nx - 1, ny - 1, nz - 1))
    with pytest.raises(L.RemoraHipError, match="32-bit indexing"):
        L.check(rem.lib.rh_vert_visc_3d(rem.ctx, bx, 0, 0, huge, huge, huge, nz - 1, 60.0))
    rem.close()


@pytest.mark.parametrize("what,msg", [("nz", "at least 3 vertical levels"), ("box", "box outside domain"), ("rank", "rank_xy outside"),
                                      ("slab", "span the full vertical"), ("ndtfast", "ndtfast"), ("gls", "gls")])
def test_rh_create_rejects_bad_parameters(what, msg):
    cfg = cases.get("Seamount")
    p = make_params(cfg)
    if what == "nz":
        p.domain_hi[2] = 1; p.box_hi[2] = 1
    elif what == "box":
        p.box_hi[0] = p.domain_hi[0] + 5
    elif what == "rank":
        p.rank_xy[0] = 3
    elif what == "slab":
        p.box_hi[2] = p.domain_hi[2] - 1
    elif what == "ndtfast":
        p.ndtfast = 0
    elif what == "gls":
        p.vmix_mode = L.VMIX_GLS                 # gls block left at zero: gls_n = 0, cmu0 = 0
    ctx = C.c_void_p()
    with pytest.raises(L.RemoraHipError, match=msg):
        L.check(L.load().rh_create(C.byref(p), C.c_void_p(0), C.byref(ctx)))


def test_gls_entry_points_need_a_gls_context():
    rem = REMORA(cases.get("Seamount"))
    with pytest.raises(L.RemoraHipError, match="RH_VMIX_GLS"):
        L.check(rem.lib.rh_gls_corrector(rem.ctx, C.byref(rem.state), rem.state.z_w, 0, 1))
    with pytest.raises(L.RemoraHipError, match="RH_VMIX_GLS"):
        L.check(rem.lib.rh_gls_prestep(rem.ctx, C.byref(rem.state), rem.state.z_w, 0, 1, 0, 0))
    rem.close()
    gl = REMORA(cases.get("Upwelling_GLS"))
    with pytest.raises(L.RemoraHipError, match="nstp/nnew"):
        L.check(gl.lib.rh_gls_corrector(gl.ctx, C.byref(gl.state), gl.state.z_w, 1, 1))
    gl.close()
