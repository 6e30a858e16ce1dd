"""Host-side mirror of the slice of `class REMORA` that drives the TimeIntegration hot path.

Names follow the reference (Source/REMORA.H:357-451, Source/REMORA.cpp:139-212, TimeIntegration/REMORA_TimeStep.cpp:9-66):
`InitData`, `Evolve`, `timeStep`, `Advance`, `setup_step`, `advance_2d_onestep`, `advance_3d_ml`.  All numerics run in the HIP
library behind include/remora_hip.h; torch only owns the device buffer (one block per rank, carved by rh_state_bind) and the
process group.  There is no CPU fallback: without a GPU and the built library every entry point raises.
"""
import ctypes as C
import json

import numpy as np

from . import lib as L
from .cases import PROB


def make_params(case, box=None, nranks=(1, 1), rank_xy=(0, 0)):
    """rh_params from a case dict (remora_amd/cases.py)."""
    p = L.Params()
    nx, ny, nz = case["ncell"]
    for d, n in enumerate((nx, ny, nz)):
        p.domain_lo[d] = 0
        p.domain_hi[d] = n - 1
    lo, hi = box if box is not None else ((0, 0, 0), (nx - 1, ny - 1, nz - 1))
    for d in range(3):
        p.box_lo[d] = lo[d]
        p.box_hi[d] = hi[d]
        p.prob_lo[d] = case["prob_lo"][d]
        p.prob_hi[d] = case["prob_hi"][d]
    for d in range(2):
        p.periodic[d] = case["periodic"][d]
        p.nranks[d] = nranks[d]
        p.rank_xy[d] = rank_xy[d]
    p.dt = case["fixed_dt"]
    p.ndtfast = case["ndtfast"]
    for k in ("flat_bathymetry", "tracer_c4", "uv_c2", "use_coriolis", "use_prestep", "use_uv3dmix", "use_barotropic", "use_salt",
              "g", "rho0", "R0", "T0", "S0", "Tcoef", "Scoef", "theta_s", "theta_b", "tcline"):
        setattr(p, k, case[k])
    prob = case["prob"]
    # init_custom_vmix / init_custom_smflux of Exec/<case>/prob.cpp
    if "gls" in case:                                        # remora.vertical_mixing_type = gls
        from .cases import GLS_STAB
        p.vmix_mode = L.VMIX_GLS
        for k, v in case["gls"].items():
            if k == "stab":
                p.gls.stability = GLS_STAB[v]
            elif k in ("p", "m", "n", "Kmin", "Pmin", "cmu0", "c1", "c2", "c3m", "c3p", "sigk", "sigp"):
                setattr(p.gls, "gls_" + k, v)
            else:
                setattr(p.gls, k, v)
    elif prob in ("Upwelling", "Advection", "DoublyPeriodic"):
        p.vmix_mode = L.VMIX_UPWELLING
    else:
        p.vmix_mode = L.VMIX_CONST
        akv, akt = {"Seamount": (1.0e-5, 1.0e-6), "DoubleGyre": (1.0, 1.0)}[prob]
        p.Akv_const = akv
        p.Akt_const[0] = akt
        p.Akt_const[1] = akt
        p.Akt_const[2] = 0.0
    p.smflux_mode = {"Upwelling": L.SMFLUX_UPWELLING, "Seamount": L.SMFLUX_NONE, "Advection": L.SMFLUX_NONE,
                     "DoubleGyre": L.SMFLUX_FIXED, "DoublyPeriodic": L.SMFLUX_UPWELLING, "Channel_Test": L.SMFLUX_CHANNEL}[prob]
    p.masks_all_one = int(case.get("masks_all_one", 1))      # every pinned case has mask == 1 (make_new_level.cpp:381-389)
    return p


def make_init_params(case):
    ip = L.InitParams()
    ip.prob = PROB[case["prob"]]
    ip.rdrag = case["rdrag"]
    ip.coriolis_f0 = case["coriolis_f0"]
    ip.coriolis_beta = case["coriolis_beta"]
    ip.hmix_constant = case["hmix_constant"]
    ip.visc2 = case["visc2"]
    for n in range(L.NCONS):
        ip.tnu2[n] = case["tnu2"][n]
    ip.u_0 = case["u_0"]
    ip.v_0 = case["v_0"]
    ip.z0 = case["z0"]
    ip.zRef = case["zRef"]
    ip.uRef = case["uRef"]
    return ip


class REMORA:
    """One level, one box per rank (max_level = 0), fixed dt: the configuration of every pinned case."""

    def __init__(self, case, device="cuda:0", box=None, nranks=(1, 1), rank_xy=(0, 0), stream=None):
        import torch
        if not torch.cuda.is_available():
            raise L.RemoraHipError("REMORA needs a HIP device; this package has no CPU path")
        self.torch = torch
        self.lib = L.load()
        self.case = case
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.params = make_params(case, box, nranks, rank_xy)
        self.ctx = C.c_void_p()
        self.stream = stream
        L.check(self.lib.rh_create(C.byref(self.params), C.c_void_p(stream.cuda_stream if stream is not None else 0), C.byref(self.ctx)))
        # bc.<var>.type / bc.<side>.type of the inputs file (REMORA::init_bcs): everything that is not periodic / SlipWall
        if "bc" in case:
            from .cases import PBC, BC_VARS
            for iv, var in enumerate(BC_VARS):
                for face, word in case["bc"][var].items():
                    L.check(self.lib.rh_set_bc(self.ctx, iv, face, PBC[word], float(case.get("bc_val", {}).get(var, {}).get(face, 0.0))))
        nbytes = self.lib.rh_state_bytes(self.ctx)
        self.buf = torch.zeros(nbytes // 8 + 32, dtype=torch.float64, device=self.device)
        base = (self.buf.data_ptr() + 255) // 256 * 256
        self._base_off = (base - self.buf.data_ptr()) // 8
        self.state = L.State()
        L.check(self.lib.rh_state_bind(self.ctx, C.c_void_p(base), C.byref(self.state)))
        self.istep = 0
        self.dt = float(case["fixed_dt"])
        self.t_new = 0.0
        self.t_old = -1.0e200
        self.nfast = self.lib.rh_nfast(self.ctx)
        self.ncell = tuple(case["ncell"])
        torch.cuda.synchronize(self.device)

    def rebind(self, buf=None):
        """Move the state to another allocation (a fresh one, or `buf`, a tensor this object made earlier): the context, its
        decomposition and its communicator stay, the arrays are bound at the new place and hold whatever the block holds (call
        InitData or upload a state next).  The step's speed depends on where the block lands (Placement); bench.py uses
        this to try a few blocks per rank without rebuilding the communicator.  Returns the previous tensor."""
        torch = self.torch
        self.sync()
        old = self.buf
        if buf is None:
            buf = torch.zeros(self.lib.rh_state_bytes(self.ctx) // 8 + 32, dtype=torch.float64, device=self.device)
        self.buf = buf
        base = (buf.data_ptr() + 255) // 256 * 256
        self._base_off = (base - buf.data_ptr()) // 8
        L.check(self.lib.rh_state_bind(self.ctx, C.c_void_p(base), C.byref(self.state)))
        self.istep = 0
        self.t_new = 0.0
        self.t_old = -1.0e200
        torch.cuda.synchronize(self.device)
        return old

    def close(self):
        if getattr(self, "ctx", None) is not None and self.ctx:
            self.lib.rh_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- communicator of a j-slab decomposed run ---------------------------------------------------------------
    def comm_init(self, dist):
        """Built-in RCCL transport: rank 0 creates the id, torch.distributed carries the 128 bytes to everyone."""
        torch = self.torch
        rank, world = dist.get_rank(), dist.get_world_size()
        idbuf = (C.c_ubyte * 128)()
        if rank == 0:
            L.check(self.lib.rh_comm_unique_id(idbuf))
        t = torch.tensor(list(idbuf), dtype=torch.uint8)
        if dist.get_backend() == "nccl":
            t = t.to(self.device)
        dist.broadcast(t, src=0)
        ids = bytes(t.cpu().tolist())
        L.check(self.lib.rh_comm_init(self.ctx, C.c_char_p(ids), rank, world))

    def comm_set_host_transport(self, dist):
        """Transport hook that stages every message through the host and torch.distributed send/recv (any backend, e.g.
        gloo).  For tests of the decomposition on boxes without one GPU per rank; RCCL is the production transport."""
        hip = C.CDLL("libamdhip64.so")
        # copies go through the stream the library hands over (hipMemcpy would order itself behind the compute stream's kernels when
        # that is the null stream, and the exchange could never run under them)
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        torch = self.torch
        self.hook_streams = []                                          # the stream of every hook call (tests: side stream or compute stream)

        def xfn(user, s_lo, r_lo, n_lo, p_lo, s_hi, r_hi, n_hi, p_hi, stream):
            try:
                self.hook_streams.append(stream or 0)
                reqs, recvs, sends = [], [], []
                for sp, n, peer, tag in ((s_lo, n_lo, p_lo, 0), (s_hi, n_hi, p_hi, 1)):      # tag 0: travels down, 1: up
                    if peer >= 0 and n > 0:
                        h = torch.empty(n, dtype=torch.float64)
                        if hip.hipMemcpyAsync(h.data_ptr(), sp, n * 8, 2, stream) != 0:
                            return 3
                        sends.append((h, peer, tag))
                if hip.hipStreamSynchronize(stream) != 0:
                    return 2
                for h, peer, tag in sends:
                    reqs.append(dist.isend(h, dst=peer, tag=tag))
                for rp, n, peer, tag in ((r_hi, n_hi, p_hi, 0), (r_lo, n_lo, p_lo, 1)):      # from above arrives what went down
                    if peer >= 0 and n > 0:
                        h = torch.empty(n, dtype=torch.float64)
                        reqs.append(dist.irecv(h, src=peer, tag=tag))
                        recvs.append((rp, h, n))
                for q in reqs:
                    q.wait()
                for rp, h, n in recvs:
                    if hip.hipMemcpyAsync(rp, h.data_ptr(), n * 8, 1, stream) != 0:
                        return 4
                if recvs and hip.hipStreamSynchronize(stream) != 0:          # the host buffers go away on return
                    return 2
                return 0
            except Exception as e:                                      # never let an exception cross the C boundary
                print("host transport failed:", repr(e), flush=True)
                return 1

        self._xfn = L.EXCHANGE_FN(xfn)                                  # keep the callback object alive
        L.check(self.lib.rh_comm_set_exchange(self.ctx, self._xfn, None, dist.get_rank(), dist.get_world_size()))

    # ---- field access (tests, diagnostics) ------------------------------------------------------------------
    def view(self, name):
        """torch view [comp, k, j, i] on the device array of a state field, plus its (lo, hi)."""
        a = getattr(self.state, name)
        off = (a.p - self.buf.data_ptr()) // 8
        return self.buf[off:off + a.size].view(a.shape), tuple(a.lo), tuple(a.hi)

    def get(self, name):
        v, lo, hi = self.view(name)
        return v.cpu().numpy(), lo, hi

    def set(self, name, arr, lo=None):
        """Upload a host array [comp, k, j, i]; `lo` = index of arr[...,0,0,0] (default: the device array's lo)."""
        v, vlo, vhi = self.view(name)
        t = self.torch.as_tensor(np.ascontiguousarray(arr), dtype=self.torch.float64)
        if lo is None:
            lo = vlo
        o = [lo[d] - vlo[d] for d in range(3)]
        nc, nk, nj, ni = t.shape
        v[:nc, o[2]:o[2] + nk, o[1]:o[1] + nj, o[0]:o[0] + ni] = t.to(self.device)

    # ---- REMORA call surface ----------------------------------------------------------------------------------
    def InitData(self):
        """REMORA::InitData for the analytic problem of the case (Source/REMORA.cpp:246-342)."""
        ip = make_init_params(self.case)
        L.check(self.lib.rh_init_state(self.ctx, C.byref(self.state), C.byref(ip)))

    def update_mskp(self):
        """REMORA::update_mskp + the periodic ghost fill of the four masks (make_new_level.cpp:617-655, init_from_netcdf.cpp:408-412):
        call after writing mskr / msku / mskv into the state (the case needs masks_all_one = 0 for the step to read them)."""
        L.check(self.lib.rh_update_mskp(self.ctx, C.byref(self.state)))

    def begin_step(self):
        """Head of REMORA::timeStep (REMORA_TimeStep.cpp:48-61): advance the clock, swap old and new."""
        self.t_old = self.t_new
        self.t_new += self.dt
        self.lib.rh_swap_old_new(C.byref(self.state))

    def end_step(self):
        self.istep += 1

    def setup_step(self):
        L.check(self.lib.rh_setup_step(self.ctx, C.byref(self.state), self.istep, self.t_old))

    def advance_2d_onestep(self, my_iif):
        L.check(self.lib.rh_advance_2d_onestep(self.ctx, C.byref(self.state), self.istep, my_iif))

    def advance_3d_ml(self):
        L.check(self.lib.rh_advance_3d_ml(self.ctx, C.byref(self.state), self.istep))

    def Advance(self):
        """REMORA::Advance (REMORA_Advance.cpp:7-35)."""
        L.check(self.lib.rh_advance(self.ctx, C.byref(self.state), self.istep, self.t_old))

    def timeStep(self):
        self.begin_step()
        self.Advance()
        self.end_step()

    def Evolve(self, nsteps):
        for _ in range(nsteps):
            self.timeStep()

    def sync(self):
        L.check(self.lib.rh_sync(self.ctx))

    def plot(self, with_scalar=False):
        """Plotfile variables [comp, k, j, i] on the valid box (IO/REMORA_Plotfile.cpp:185-235)."""
        lo = [self.params.box_lo[d] for d in range(3)]
        hi = [self.params.box_hi[d] for d in range(3)]
        shp = (6 if with_scalar else 5, hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1)
        out = self.torch.empty(shp, dtype=self.torch.float64, device=self.device)
        L.check(self.lib.rh_plot_vars(self.ctx, C.byref(self.state), int(with_scalar), C.c_void_p(out.data_ptr())))
        self.sync()
        return out.cpu().numpy()

    def WritePlotFile(self, dirname, with_scalar=False):
        """REMORA::WritePlotFile (IO/REMORA_Plotfile.cpp:141-433): AMReX-native plotfile of the current state (single box)."""
        L.check(self.lib.rh_write_plotfile(self.ctx, C.byref(self.state), str(dirname).encode(), int(with_scalar), self.istep, self.t_new))

    def WriteCheckpointFile(self, dirname):
        """REMORA::WriteCheckpointFile (IO/REMORA_Checkpoint.cpp:15-245)."""
        L.check(self.lib.rh_write_checkpoint(self.ctx, C.byref(self.state), str(dirname).encode(), self.istep, self.t_new))

    def ReadCheckpointFile(self, dirname):
        """REMORA::ReadCheckpointFile + restart (IO/REMORA_Checkpoint.cpp:248-470): InitData for the arrays no checkpoint
        holds, then the checkpointed arrays, the step counter and the clock."""
        self.InitData()
        st, t = C.c_int(), C.c_double()
        L.check(self.lib.rh_read_checkpoint(self.ctx, C.byref(self.state), str(dirname).encode(), C.byref(st), C.byref(t)))
        self.istep = st.value
        self.t_new = t.value
        self.t_old = t.value - self.dt

    def write_plot_data(self, dirname, file_index, with_scalar=False):
        """This rank's FAB records of a decomposed run; returns the PlotBox record for rh_plotfile_write_headers."""
        pb = L.PlotBox()
        L.check(self.lib.rh_plotfile_write_data(self.ctx, C.byref(self.state), str(dirname).encode(), int(with_scalar), file_index, C.byref(pb)))
        return pb

    # ---- reductions around the step ------------------------------------------------------------------------
    def estTimeStep(self, cfl=0.8, local_only=False):
        """REMORA::estTimeStep (REMORA_ComputeTimestep.cpp:39-95)."""
        dt = C.c_double()
        L.check(self.lib.rh_est_timestep(self.ctx, C.byref(self.state), cfl, int(local_only), C.byref(dt)))
        return dt.value

    def sum_integrated_quantities(self, local_only=False):
        """REMORA::sum_integrated_quantities (REMORA_SumIQ.cpp:9-106): (scalar, kineng, volume, max_vel)."""
        out = (C.c_double * 4)()
        L.check(self.lib.rh_sum_integrated_quantities(self.ctx, C.byref(self.state), int(local_only), out))
        return tuple(out)

    def build_id(self):
        return self.lib.rh_build_id().decode()

    # ---- per-kernel timing ---------------------------------------------------------------------------------
    def timing(self, on=True):
        L.check(self.lib.rh_timing_enable(self.ctx, int(on)))

    def timing_reset(self):
        L.check(self.lib.rh_timing_reset(self.ctx))

    def timing_report(self):
        buf = C.create_string_buffer(1 << 16)
        L.check(self.lib.rh_timing_report(self.ctx, buf, len(buf)))
        return json.loads(buf.value.decode())
