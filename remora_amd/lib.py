"""ctypes binding of the C-ABI library `libremora_hip.so` (declared in include/remora_hip.h).

The library is the product: there is no CPU path behind it.  If it is missing, or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libremora_hip.so")

NCONS = 3
NGROW = 2

# BCVars / modes (include/remora_hip.h)
VMIX_CONST, VMIX_UPWELLING, VMIX_GLS = 0, 1, 2
SMFLUX_NONE, SMFLUX_UPWELLING, SMFLUX_FIXED, SMFLUX_CHANNEL = 0, 1, 2, 3


class Array4(C.Structure):
    _fields_ = [("p", C.c_void_p), ("lo", C.c_int * 3), ("hi", C.c_int * 3), ("ncomp", C.c_int)]

    @property
    def shape(self):
        """numpy/torch shape [comp, k, j, i] of the array (i fastest in memory)."""
        return (self.ncomp, self.hi[2] - self.lo[2] + 1, self.hi[1] - self.lo[1] + 1, self.hi[0] - self.lo[0] + 1)

    @property
    def size(self):
        s = self.shape
        return s[0] * s[1] * s[2] * s[3]


class Box(C.Structure):
    _fields_ = [("lo", C.c_int * 3), ("hi", C.c_int * 3)]


def box(lo, hi):
    b = Box()
    for d in range(3):
        b.lo[d] = lo[d]
        b.hi[d] = hi[d]
    return b


class GlsParams(C.Structure):
    _fields_ = [("stability", C.c_int)] + [(k, C.c_double) for k in (
        "gls_p", "gls_m", "gls_n", "gls_Kmin", "gls_Pmin", "gls_cmu0", "gls_c1", "gls_c2", "gls_c3m", "gls_c3p", "gls_sigk",
        "gls_sigp", "Akk_bak", "Akp_bak", "Akv_bak", "Akt_bak", "Zos", "Zob")]


class Params(C.Structure):
    _fields_ = [("domain_lo", C.c_int * 3), ("domain_hi", C.c_int * 3), ("periodic", C.c_int * 2),
                ("box_lo", C.c_int * 3), ("box_hi", C.c_int * 3), ("nranks", C.c_int * 2), ("rank_xy", C.c_int * 2),
                ("prob_lo", C.c_double * 3), ("prob_hi", C.c_double * 3), ("dt", C.c_double), ("ndtfast", C.c_int),
                ("flat_bathymetry", C.c_int), ("tracer_c4", C.c_int), ("uv_c2", C.c_int), ("use_coriolis", C.c_int),
                ("use_prestep", C.c_int), ("use_uv3dmix", C.c_int), ("use_barotropic", C.c_int), ("use_salt", C.c_int),
                ("g", C.c_double), ("rho0", C.c_double), ("R0", C.c_double), ("T0", C.c_double), ("S0", C.c_double),
                ("Tcoef", C.c_double), ("Scoef", C.c_double), ("theta_s", C.c_double), ("theta_b", C.c_double),
                ("tcline", C.c_double), ("vmix_mode", C.c_int), ("Akv_const", C.c_double), ("Akt_const", C.c_double * NCONS),
                ("smflux_mode", C.c_int), ("masks_all_one", C.c_int), ("gls", GlsParams)]


STATE_FIELDS = ["cons_old", "cons_new", "xvel_old", "xvel_new", "yvel_old", "yvel_new", "zvel_old", "zvel_new",
                "Zt_avg1", "h", "ubar", "vbar", "ru", "rv", "ru2d", "rv2d", "mskr", "msku", "mskv", "mskp",
                "z_w", "z_r", "Hz", "Huon", "Hvom", "Akv", "Akt", "visc2_p", "visc2_r", "diff2",
                "rufrc", "rvfrc", "sustr", "svstr", "rdrag", "bustr", "bvstr", "DU_avg1", "DU_avg2", "DV_avg1", "DV_avg2",
                "rubar", "rvbar", "rzeta", "zeta", "pm", "pn", "fcor", "sstore", "rhoS", "rhoA", "bvf",
                "tke", "gls", "Lscale", "Akk", "Akp"]


class State(C.Structure):
    _fields_ = [(n, Array4) for n in STATE_FIELDS]


class InitParams(C.Structure):
    _fields_ = [("prob", C.c_int), ("rdrag", C.c_double), ("coriolis_f0", C.c_double), ("coriolis_beta", C.c_double),
                ("hmix_constant", C.c_int), ("visc2", C.c_double), ("tnu2", C.c_double * NCONS), ("u_0", C.c_double), ("v_0", C.c_double),
                ("z0", C.c_double), ("zRef", C.c_double), ("uRef", C.c_double)]


class HaloPlan(C.Structure):
    _fields_ = [("peer_lo", C.c_int), ("peer_hi", C.c_int), ("nrows", C.c_int), ("i_lo", C.c_int), ("i_hi", C.c_int),
                ("send_lo_j", C.c_int), ("send_hi_j", C.c_int), ("recv_lo_j", C.c_int), ("recv_hi_j", C.c_int)]


# rh_exchange_fn (include/remora_hip.h)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p)


class PlotBox(C.Structure):
    _fields_ = [("box", Box), ("file_index", C.c_int), ("cell_offset", C.c_long), ("nd_offset", C.c_long),
                ("cell_min", C.c_double * 6), ("cell_max", C.c_double * 6), ("nd_min", C.c_double * 3), ("nd_max", C.c_double * 3)]


class RemoraHipError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol include/remora_hip.h declares
_P = C.c_void_p
_A = Array4
_SP = C.POINTER(State)
SIGNATURES = {
    "rh_last_error": (C.c_char_p, []),
    "rh_build_id": (C.c_char_p, []),
    "rh_create": (C.c_int, [C.POINTER(Params), _P, C.POINTER(_P)]),
    "rh_destroy": (C.c_int, [_P]),
    "rh_nfast": (C.c_int, [_P]),
    "rh_weights": (C.c_int, [_P, _P, _P]),
    "rh_sync": (C.c_int, [_P]),
    "rh_state_bytes": (C.c_long, [_P]),
    "rh_state_bind": (C.c_int, [_P, _P, _SP]),
    "rh_comm_unique_id": (C.c_int, [_P]),
    "rh_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "rh_comm_set_exchange": (C.c_int, [_P, EXCHANGE_FN, _P, C.c_int, C.c_int]),
    "rh_comm_selftest": (C.c_int, [_P, C.c_long]),
    "rh_halo_plan_j": (C.c_int, [C.POINTER(Params), C.c_int * 3, C.c_int * 3, C.c_int * 3, C.POINTER(HaloPlan)]),
    "rh_est_timestep": (C.c_int, [C.c_void_p, C.POINTER(State), C.c_double, C.c_int, C.POINTER(C.c_double)]),
    "rh_sum_integrated_quantities": (C.c_int, [C.c_void_p, C.POINTER(State), C.c_int, C.c_double * 4]),
    "rh_sum_integrated_quantities_noalloc": (C.c_int, [C.c_void_p, C.POINTER(State), C.c_void_p]),
    "rh_halo_plan_i": (C.c_int, [C.POINTER(Params), C.c_int * 3, C.c_int * 3, C.c_int * 3, C.POINTER(HaloPlan)]),
    "rh_fill_boundary": (C.c_int, [_P, _A, C.c_int * 3, C.c_int, C.c_int]),
    "rh_physbc": (C.c_int, [_P, _A, C.c_int * 3, _A, C.c_int, C.c_int, C.c_int]),
    "rh_fill_patch": (C.c_int, [_P, _A, C.c_int * 3, _A, C.c_int, C.c_int, C.c_int]),
    "rh_physbc_calc": (C.c_int, [_P, _A, C.c_int * 3, _A, C.c_int, C.c_int, C.c_int, C.c_int, _A, _A, _A]),
    "rh_fill_patch_calc": (C.c_int, [_P, _A, C.c_int * 3, _A, C.c_int, C.c_int, C.c_int, _A, _A, _A]),
    "rh_set_bc": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_double]),
    "rh_get_bc": (C.c_int, [_P, C.c_int, C.c_int * 6]),
    "rh_stretch_transform": (C.c_int, [_P, _SP]),
    "rh_rho_eos": (C.c_int, [_P, Box] + [_A] * 10 + [C.c_int]),
    "rh_prsgrd": (C.c_int, [_P, Box, Box, Box] + [_A] * 11 + [C.c_int, C.c_int]),
    "rh_omega": (C.c_int, [_P, Box] + [_A] * 5 + [C.c_int]),
    "rh_prestep_t": (C.c_int, [_P, Box] + [_A] * 11 + [C.c_int, C.c_int, C.c_int, C.c_double]),
    "rh_prestep_uv": (C.c_int, [_P, Box, C.c_int, C.c_int] + [_A] * 8 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
    "rh_t3dmix": (C.c_int, [_P, Box] + [_A] * 8 + [C.c_double, C.c_int]),
    "rh_plotfile_write_data": (C.c_int, [_P, C.POINTER(State), C.c_char_p, C.c_int, C.c_int, C.POINTER(PlotBox)]),
    "rh_plotfile_write_headers": (C.c_int, [C.POINTER(Params), C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(PlotBox)]),
    "rh_write_plotfile": (C.c_int, [_P, C.POINTER(State), C.c_char_p, C.c_int, C.c_int, C.c_double]),
    "rh_write_checkpoint": (C.c_int, [_P, C.POINTER(State), C.c_char_p, C.c_int, C.c_double]),
    "rh_read_checkpoint": (C.c_int, [_P, C.POINTER(State), C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "rh_gls_prestep": (C.c_int, [_P, C.POINTER(State), _A, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rh_gls_corrector": (C.c_int, [_P, C.POINTER(State), _A, C.c_int, C.c_int]),
    "rh_coriolis": (C.c_int, [_P, Box, Box] + [_A] * 6 + [C.c_int, C.c_int]),
    "rh_rhs_uv_3d": (C.c_int, [_P, Box, Box] + [_A] * 17 + [C.c_int, C.c_int, C.c_int]),
    "rh_uv3dmix": (C.c_int, [_P, Box, Box] + [_A] * 12 + [C.c_int, C.c_int, C.c_double]),
    "rh_vert_visc_3d": (C.c_int, [_P, Box, C.c_int, C.c_int, _A, _A, _A, C.c_int, C.c_double]),
    "rh_vert_mean_3d": (C.c_int, [_P, Box, C.c_int, C.c_int] + [_A] * 5 + [C.c_int]),
    "rh_update_massflux_3d": (C.c_int, [_P, Box, C.c_int, C.c_int] + [_A] * 8 + [C.c_int]),
    "rh_rhs_t_3d": (C.c_int, [_P, Box] + [_A] * 10 + [C.c_int, C.c_double]),
    "rh_rhs_uv_2d": (C.c_int, [_P, Box, Box] + [_A] * 6 + [C.c_int]),
    "rh_setup_step": (C.c_int, [_P, _SP, C.c_int, C.c_double]),
    "rh_advance_2d": (C.c_int, [_P, _SP, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rh_advance_2d_onestep": (C.c_int, [_P, _SP, C.c_int, C.c_int]),
    "rh_advance_3d_ml": (C.c_int, [_P, _SP, C.c_int]),
    "rh_advance": (C.c_int, [_P, _SP, C.c_int, C.c_double]),
    "rh_swap_old_new": (None, [_SP]),
    "rh_init_state": (C.c_int, [_P, _SP, C.POINTER(InitParams)]),
    "rh_plot_vars": (C.c_int, [_P, _SP, C.c_int, _P]),
    "rh_update_mskp": (C.c_int, [_P, _SP]),
    "rh_timing_enable": (C.c_int, [_P, C.c_int]),
    "rh_timing_report": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "rh_timing_reset": (C.c_int, [_P]),
}


def load():
    """Load the library and attach signatures.  Raises if the library or any declared symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RemoraHipError(f"{LIB_PATH} is missing: build it with `make -C remora_amd/csrc` "
                             "(or __graft_entry__.build()); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)          # AttributeError if the symbol is not exported
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RemoraHipError(load().rh_last_error().decode(errors="replace"))
