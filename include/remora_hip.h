/*
 * remora_hip.h -- C ABI of the MI355X-native replacement for REMORA's TimeIntegration hot path.
 *
 * REMORA has no plugin/FFI layer; the boundary this library slots into is the C++ member-function
 * surface of `class REMORA` (Source/REMORA.H:357-648).  Every entry point below names the member
 * function (file:line in the reference tree) whose body it replaces.  Conventions:
 *   - all pointers are DEVICE pointers (hipMalloc / AMReX The_Arena on a HIP build), borrowed, never freed here;
 *   - rh_array4 is amrex::Array4<Real>: p addresses element (lo[0],lo[1],lo[2],comp 0),
 *       index(i,j,k,n) = (i-lo0) + (j-lo1)*nx + (k-lo2)*nx*ny + n*nx*ny*nz,  nx = hi0-lo0+1, ...
 *     so MultiFab::array(mfi) maps 1:1 (lo/hi = grown FAB box, global index space);
 *   - arrays are updated in place exactly where the reference updates them;
 *   - return value 0 = ok; non-zero = error (rh_last_error() gives the text); the AMReX-side wrapper
 *     maps non-zero to amrex::Abort, the reference's only error channel (e.g. REMORA_rhs_t_3d.cpp:143);
 *   - work is enqueued on the stream given to rh_create (NULL = default stream); calls do not synchronise
 *     unless stated.
 * FP64 throughout, like the reference (CMakeLists.txt:17).
 */
#ifndef REMORA_HIP_H
#define REMORA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RH_NCONS 3      /* Source/REMORA_IndexDefines.H:8-11 */
#define RH_NGROW 2      /* Source/REMORA_IndexDefines.H:13 */

typedef struct { double *p; int lo[3]; int hi[3]; int ncomp; } rh_array4;
typedef struct { int lo[3]; int hi[3]; } rh_box;

/* mathematical boundary types, Source/REMORA_IndexDefines.H:61-73 */
enum { RH_BC_REFLECT_ODD = -1, RH_BC_INT_DIR = 0, RH_BC_REFLECT_EVEN = 1, RH_BC_FOEXTRAP = 2, RH_BC_EXT_DIR = 3, RH_BC_CLAMPED = 4,
       RH_BC_CHAPMAN = 5, RH_BC_FLATHER = 6, RH_BC_ORLANSKI_RAD = 7, RH_BC_ORLANSKI_RAD_NUDGE = 8 };
/* physical boundary kinds: enum struct REMORA_BC, Source/REMORA_IndexDefines.H:53-56 (the words of bc.<side|var>.type) */
enum { RH_PBC_SYMMETRY = 0, RH_PBC_INFLOW, RH_PBC_OUTFLOW, RH_PBC_NO_SLIP_WALL, RH_PBC_SLIP_WALL, RH_PBC_PERIODIC, RH_PBC_CLAMPED,
       RH_PBC_CHAPMAN, RH_PBC_FLATHER, RH_PBC_ORLANSKI_RAD, RH_PBC_ORLANSKI_RAD_NUDGE, RH_PBC_UNDEFINED };
/* BCVars, Source/REMORA_IndexDefines.H:15-34 */
enum { RH_BCV_CONS = 0, RH_BCV_XVEL = 3, RH_BCV_YVEL, RH_BCV_ZVEL, RH_BCV_UBAR, RH_BCV_VBAR, RH_BCV_ZETA, RH_BCV_TKE,
       RH_BCV_FOEXTRAP_PERIODIC, RH_BCV_FOEXTRAP, RH_BCV_U2D_SIMPLE, RH_BCV_V2D_SIMPLE, RH_BCV_NUM };
/* analytic vertical mixing profiles = the init_custom_vmix hooks of Exec/<case>/prob.cpp that set_analytical_vmix
 * (Source/REMORA.cpp:565-573) calls from setup_step every step */
/* RH_VMIX_CONST: the reference rewrites the same constants into Akv / Akt at every step.  Here the drivers (rh_setup_step,
 * rh_advance) write them into the bound Akv / Akt once -- at the first step after rh_state_bind, rh_init_state or rh_read_checkpoint --
 * and from then on give the column solvers the constants themselves instead of the arrays: in this mode the library owns the contents
 * of Akv / Akt from that first step on; a caller that writes other values into them must bind the state again (or run with
 * RH_VMIX_REWRITE=1: the reference's per-step rewrite and array reads).  The leaf entry point rh_vert_visc_3d always reads the
 * array it is given. */
enum { RH_VMIX_CONST = 0, RH_VMIX_UPWELLING = 1,
       RH_VMIX_GLS = 2 };   /* remora.vertical_mixing_type = gls: Akv / Akt come from the closure (rh_gls_*), not from a profile */
/* remora.gls_stability_type, Source/REMORA_DataStruct.H:206-224 */
enum { RH_GLS_CANUTO_A = 0, RH_GLS_CANUTO_B = 1, RH_GLS_GALPERIN = 2 };
/* SolverChoice's GLS block, Source/REMORA_DataStruct.H:203-204,228-268,399-400,427-445 (read only when vmix_mode = RH_VMIX_GLS) */
typedef struct rh_gls_params {
    int    stability;
    double gls_p, gls_m, gls_n, gls_Kmin, gls_Pmin, gls_cmu0, gls_c1, gls_c2, gls_c3m, gls_c3p, gls_sigk, gls_sigp;
    double Akk_bak, Akp_bak, Akv_bak, Akt_bak, Zos, Zob;
} rh_gls_params;
/* analytic surface momentum flux = init_custom_smflux hooks called by set_smflux (Source/REMORA.cpp:598-605) */
enum { RH_SMFLUX_NONE = 0, RH_SMFLUX_UPWELLING = 1, RH_SMFLUX_FIXED = 2,
       RH_SMFLUX_CHANNEL = 3 };   /* Exec/Channel_Test/prob.cpp:204-222: uniform along-channel wind oscillating at 0.9 f */

/* Hidden inputs of the reference's member functions: geom[0] (domain, periodicity), solverChoice
 * (Source/REMORA_DataStruct.H:351-483), fixed_ndtfast_ratio, and the process grid of this rank. */
typedef struct rh_params {
    int    domain_lo[3], domain_hi[3];   /* cell-centred domain box, geom[0].Domain() */
    int    periodic[2];                  /* geometry.is_periodic (x,y) */
    int    box_lo[3], box_hi[3];         /* this rank's valid cell box (full vertical extent) */
    int    nranks[2], rank_xy[2];        /* i-j process grid and this rank's coordinates in it */
    double prob_lo[3], prob_hi[3];
    double dt;                           /* remora.fixed_dt */
    int    ndtfast;                      /* remora.fixed_ndtfast_ratio */
    int    flat_bathymetry, tracer_c4, uv_c2, use_coriolis, use_prestep, use_uv3dmix, use_barotropic, use_salt;
    double g, rho0, R0, T0, S0, Tcoef, Scoef, theta_s, theta_b, tcline;
    int    vmix_mode;  double Akv_const, Akt_const[RH_NCONS];
    int    smflux_mode;
    int    masks_all_one;                /* 1: land/sea masks are identically 1 (all pinned cases); skips the final cons*=mskr pass */
    rh_gls_params gls;
} rh_params;

/* One view per MultiFab the reference's drivers touch (Source/REMORA.H:189-354; shapes and ghost widths:
 * Source/Initialization/REMORA_make_new_level.cpp:259-282,381-389,426-503).  z ghost cells are optional. */
typedef struct rh_state {
    rh_array4 cons_old, cons_new, xvel_old, xvel_new, yvel_old, yvel_new, zvel_old, zvel_new;
    rh_array4 Zt_avg1, h, ubar, vbar, ru, rv, ru2d, rv2d;
    rh_array4 mskr, msku, mskv, mskp;
    rh_array4 z_w, z_r, Hz, Huon, Hvom, Akv, Akt, visc2_p, visc2_r, diff2;
    rh_array4 rufrc, rvfrc, sustr, svstr, rdrag, bustr, bvstr, DU_avg1, DU_avg2, DV_avg1, DV_avg2;
    rh_array4 rubar, rvbar, rzeta, zeta, pm, pn, fcor, sstore, rhoS, rhoA, bvf;
    /* GLS closure (make_new_level.cpp:499-503), w-points, ghost width 2: tke, gls x3 components (two time levels + predictor),
     * Lscale, Akk, Akp.  Bound (and counted by rh_state_bytes) only when vmix_mode = RH_VMIX_GLS; null views otherwise. */
    rh_array4 tke, gls, Lscale, Akk, Akp;
} rh_state;

typedef struct rh_ctx rh_ctx;

const char *rh_last_error(void);
/* Build tag of the library (bumped whenever a kernel changes): profiles/traffic_current.json records the tag of the build its PMC
 * passes ran on, and bench.py reports those bytes only for the same build.  No counterpart in the reference. */
const char *rh_build_id(void);

/* ---- context ------------------------------------------------------------------------------------- */
/* Builds the context: BC table (REMORA::init_bcs, Initialization/REMORA_init_bcs.cpp:9-679; every non-periodic side starts as
 * SlipWall, rh_set_bc changes that),
 * barotropic filter weights (REMORA::set_weights, TimeIntegration/REMORA_set_weights.cpp:10-171), stretching curves
 * (k-only part of Utils/REMORA_DepthStretchTransform.H:81-139) and device work space.  stream: hipStream_t or NULL. */
int  rh_create(const rh_params *prm, void *stream, rh_ctx **out);
int  rh_destroy(rh_ctx *ctx);
/* REMORA::init_bcs for one variable and face: what "bc.<var>.type" (remora.boundary_per_variable = 1) or "bc.<side>.type" puts
 * into phys_bc_type (Initialization/REMORA_init_bcs.cpp:9-199), followed by the translation into the per-component mathematical
 * types (:201-679), which is redone as a whole on every call.  bcvar: one of the ten user variables in BCVars order (RH_BCV_CONS+0
 * temp, +1 salt, +2 scalar, RH_BCV_XVEL u, RH_BCV_YVEL v, RH_BCV_ZVEL w, RH_BCV_UBAR, RH_BCV_VBAR, RH_BCV_ZETA, RH_BCV_TKE), or
 * -1 for all ten (the bc.<side>.type form).  face: amrex::Orientation index 0 xlo, 1 ylo, 3 xhi, 4 yhi (z faces are always
 * SlipWall, :130).  kind: RH_PBC_*.  value: bc.<>.velocity[component] / bc.<>.scalar of an inflow face (velocity of a no-slip
 * face: its normal component is forced to 0, :51), ignored otherwise.  Errors where the reference aborts: a non-periodic kind on a
 * periodic direction and vice versa, chapman on anything but zeta, flather on anything but ubar / vbar, kinds a variable's
 * translation does not accept (:307,:407,:539,:624).  clamped / chapman / flather / orlanski_rad_nudg sides are filled as the
 * reference's physbc fills them; the boundary DATA those kinds relax to comes from fill_from_bdyfiles (NetCDF), which is outside
 * this library. */
int  rh_set_bc(rh_ctx *ctx, int bcvar, int face, int kind, double value);
/* the translated table: out6 = lo(x,y,z), hi(x,y,z) mathematical types of BCVars entry bccomp (domain_bcs_type[bccomp]) */
int  rh_get_bc(const rh_ctx *ctx, int bccomp, int out6[6]);
int  rh_nfast(const rh_ctx *ctx);                                  /* REMORA::nfast (REMORA.H:937) */
int  rh_weights(const rh_ctx *ctx, double *w1, double *w2);        /* vec_weight1/2, 2*ndtfast+1 values each (host) */
int  rh_sync(rh_ctx *ctx);                                         /* hipStreamSynchronize on the context's stream */

/* Bytes of device memory a native rh_state for this rank needs, and binding of views into such a block
 * (replaces the MultiFab allocations of make_new_level.cpp:259-282,426-503 when the caller is not AMReX). */
long rh_state_bytes(const rh_ctx *ctx);
int  rh_state_bind(const rh_ctx *ctx, void *device_base, rh_state *out);

/* Optional RCCL communicator for i-j decomposed runs: rank 0 calls rh_comm_unique_id, the caller broadcasts the
 * 128 bytes (e.g. torch.distributed), every rank calls rh_comm_init.  Replaces the MPI side of FillBoundary. */
int  rh_comm_unique_id(void *id128);
int  rh_comm_init(rh_ctx *ctx, const void *id128, int rank, int nranks);

/* Transport hook: replaces the built-in RCCL send/recv pair of the ghost exchange, for hosts that own their communicator
 * (REMORA itself: AMReX ParallelDescriptor / GPU-aware MPI) and for tests (gloo).  Called once per exchange phase (X: lower =
 * left, upper = right neighbour; Y: lower / upper row neighbour; all arrays of one fill scope share the message) with DEVICE
 * buffers that are ready on `stream`: send n_lo doubles from send_lo to rank peer_lo and receive n_lo doubles from it into
 * recv_lo, likewise for the upper neighbour; a peer of -1 means no neighbour on that side.  The block sent to the lower
 * neighbour is what that neighbour receives from its upper one.  On return the receive buffers must be complete on `stream`.
 * Stream contract: ALL device work of an exchange is ordered on the `stream` argument, never on the caller's own stream -- the pack
 * kernels that wrote the send buffers were launched on it and the unpack kernels that read the receive buffers will be.  It is the
 * context's compute stream or, while an exchange runs under compute kernels (the ghost fills at the end of rh_setup_step, the
 * interior / rim split of the barotropic loop; both on by default with a hook), the library's side stream.  A hook that hands the
 * buffers to a host-driven transport (MPI) must synchronise exactly that stream first (hipStreamSynchronize((hipStream_t)stream)).
 * Return 0 on success. */
typedef int (*rh_exchange_fn)(void *user, const double *send_lo, double *recv_lo, long n_lo, int peer_lo,
                              const double *send_hi, double *recv_hi, long n_hi, int peer_hi, void *stream);
int  rh_comm_set_exchange(rh_ctx *ctx, rh_exchange_fn fn, void *user, int rank, int nranks);

/* Diagnostic of the active transport: sends two n-double patterns to this rank itself and checks the arrival order.
 * No counterpart in the reference. */
int  rh_comm_selftest(rh_ctx *ctx, long n);

/* Ghost blocks a rank exchanges with its neighbours for one array (host-only, needs no device): the geometry behind
 * rh_fill_boundary on an i-j process grid (rank = rank_xy[1] * nranks[0] + rank_xy[0]).  The exchange runs in two phases: X swaps
 * ghost columns with the left / right neighbour over the valid rows, then Y swaps ghost rows with the lower / upper neighbour
 * over the full width (ghost columns included), so corner blocks arrive in two hops and no diagonal message exists.
 * arr_lo/arr_hi = the array's grown box, ng = its ghost widths.  rh_halo_plan_j describes phase Y.  rh_halo_plan_i describes
 * phase X with the roles of i and j swapped: nrows = ghost columns, i_lo/i_hi = the ROW range of every column, *_j = first COLUMN. */
typedef struct rh_halo_plan {
    int peer_lo, peer_hi;        /* rank of the lower / upper neighbour, -1 = none (domain edge, not periodic; or a single rank) */
    int nrows;                   /* ghost rows exchanged on each side */
    int i_lo, i_hi;              /* i-range of every row sent and received (x ghost columns only where FillBoundary fills them) */
    int send_lo_j, send_hi_j;    /* first row of the block sent to the lower / to the upper neighbour */
    int recv_lo_j, recv_hi_j;    /* first row of the ghost block received from the lower / from the upper neighbour */
} rh_halo_plan;
int  rh_halo_plan_j(const rh_params *prm, const int arr_lo[3], const int arr_hi[3], const int ng[3], rh_halo_plan *out);
int  rh_halo_plan_i(const rh_params *prm, const int arr_lo[3], const int arr_hi[3], const int ng[3], rh_halo_plan *out);

/* ---- ghost cells ----------------------------------------------------------------------------------- */
/* amrex::FabArray::FillBoundary(periodicity) for comps [icomp,icomp+ncomp): periodic wrap and, across ranks,
 * neighbour exchange (call sites: SURVEY.md section 2.4).  valid = a's box shrunk by ng. */
int  rh_fill_boundary(rh_ctx *ctx, rh_array4 a, const int ng[3], int icomp, int ncomp);
/* REMORAPhysBCFunct::operator() (BoundaryConditions/REMORA_PhysBCFunct.cpp:21-93; impose_cons_bcs / impose_xvel_bcs /
 * impose_yvel_bcs / impose_zvel_bcs, BoundaryConditions/REMORA_BoundaryConditions_{cons,xvel,yvel,zvel}.cpp) with an empty mf_calc:
 * an orlanski_rad side is an error here (the reference reads an empty Array4 there) */
int  rh_physbc(rh_ctx *ctx, rh_array4 a, const int ng[3], rh_array4 msk, int icomp, int ncomp, int bccomp);
/* REMORA::FillPatch at level 0 (BoundaryConditions/REMORA_FillPatch.cpp:16-155) = rh_fill_boundary + rh_physbc */
int  rh_fill_patch(rh_ctx *ctx, rh_array4 a, const int ng[3], rh_array4 msk, int bccomp, int icomp, int ncomp);
/* The same with the arguments the radiation condition reads (PhysBCFunct.cpp:21-23): n_not_fill, mf_calc (the previous time level;
 * calc.p = NULL stands for the empty MultiFab; calc may be the filled array itself, as in REMORA::InitData, REMORA.cpp:293-295: the
 * passes then run in the order of the reference's CPU loops) and, for cell-centred arrays, msku / mskv. */
int  rh_physbc_calc(rh_ctx *ctx, rh_array4 a, const int ng[3], rh_array4 msk, int icomp, int ncomp, int bccomp, int n_not_fill,
                    rh_array4 calc, rh_array4 msku, rh_array4 mskv);
int  rh_fill_patch_calc(rh_ctx *ctx, rh_array4 a, const int ng[3], rh_array4 msk, int bccomp, int icomp, int ncomp,
                        rh_array4 calc, rh_array4 msku, rh_array4 mskv);

/* ---- leaves (Box/Array4 level, Source/REMORA.H:454-648) -------------------------------------------- */
/* REMORA::stretch_transform, Utils/REMORA_DepthStretchTransform.H:12-249 (z_w, z_r, Hz, h comp 1 from h, Zt_avg1) */
int  rh_stretch_transform(rh_ctx *ctx, rh_state *s);
/* REMORA::rho_eos, TimeIntegration/REMORA_rho_eos.cpp:22-99 */
int  rh_rho_eos(rh_ctx *ctx, rh_box bx, rh_array4 state, rh_array4 rho, rh_array4 rhoA, rh_array4 rhoS, rh_array4 bvf,
                rh_array4 Hz, rh_array4 z_w, rh_array4 z_r, rh_array4 h, rh_array4 mskr, int N);
/* REMORA::prsgrd, TimeIntegration/REMORA_prsgrd.cpp:6-197 (P: caller scratch on phi_bx) */
int  rh_prsgrd(rh_ctx *ctx, rh_box phi_bx, rh_box utbx, rh_box vtbx, rh_array4 ru, rh_array4 rv, rh_array4 pn, rh_array4 pm,
               rh_array4 rho, rh_array4 P, rh_array4 Hz, rh_array4 z_r, rh_array4 z_w, rh_array4 msku, rh_array4 mskv,
               int nrhs, int N);
/* omega: TimeIntegration/REMORA_prestep_t_advection.cpp:97-125 == REMORA_advance_3d.cpp:272-293 */
int  rh_omega(rh_ctx *ctx, rh_box gbx1, rh_array4 W, rh_array4 Huon, rh_array4 Hvom, rh_array4 z_w, rh_array4 h, int N);
/* REMORA::prestep_t_advection (REMORA_prestep_t_advection.cpp:127-361, after omega) fused with the tracer call of
 * REMORA::prestep_diffusion (REMORA_prestep_diffusion.cpp:19-165, lambda = 1): tempstore and t_new for ONE tracer. */
int  rh_prestep_t(rh_ctx *ctx, rh_box tbx, rh_array4 told, rh_array4 tnew, rh_array4 tempstore, rh_array4 Hz,
                  rh_array4 Huon, rh_array4 Hvom, rh_array4 W, rh_array4 pm, rh_array4 pn, rh_array4 msku, rh_array4 mskv,
                  int iic, int ntfirst, int N, double dt);
/* REMORA::prestep_diffusion for u (ioff=1) or v (joff=1), REMORA_prestep_diffusion.cpp:19-165 */
int  rh_prestep_uv(rh_ctx *ctx, rh_box vel_bx, int ioff, int joff, rh_array4 vel, rh_array4 vel_old, rh_array4 rvel,
                   rh_array4 Hz, rh_array4 sstr, rh_array4 bstr, rh_array4 pm, rh_array4 pn,
                   int iic, int ntfirst, int nrhs, int N, double dt);
/* REMORA::t3dmix, TimeIntegration/REMORA_t3dmix.cpp:6-58 */
int  rh_t3dmix(rh_ctx *ctx, rh_box bx, rh_array4 state, rh_array4 state_rhs, rh_array4 diff2, rh_array4 Hz, rh_array4 pm,
               rh_array4 pn, rh_array4 msku, rh_array4 mskv, double dt, int ncomp);
/* REMORA::coriolis, TimeIntegration/REMORA_coriolis.cpp:10-40, as a leaf of its own (fomn = fcor / (pm pn), REMORA_setup_step.cpp:233-239;
 * also the 2-D call of REMORA_advance_2d.cpp:530 with Hz := Drhs on k = 0 slabs).  Inside rh_rhs_uv_3d / rh_advance_2d it is fused. */
int  rh_coriolis(rh_ctx *ctx, rh_box xbx, rh_box ybx, rh_array4 uold, rh_array4 vold, rh_array4 ru, rh_array4 rv,
                 rh_array4 Hz, rh_array4 fomn, int nrhs, int nr);
/* REMORA::coriolis + REMORA::rhs_uv_3d, TimeIntegration/REMORA_coriolis.cpp:10-40, REMORA_rhs_uv_3d.cpp:31-385 */
int  rh_rhs_uv_3d(rh_ctx *ctx, rh_box xbx, rh_box ybx, rh_array4 uold, rh_array4 vold, rh_array4 ru, rh_array4 rv,
                  rh_array4 rufrc, rh_array4 rvfrc, rh_array4 sustr, rh_array4 svstr, rh_array4 bustr, rh_array4 bvstr,
                  rh_array4 Huon, rh_array4 Hvom, rh_array4 pm, rh_array4 pn, rh_array4 W, rh_array4 Hz, rh_array4 fcor,
                  int use_coriolis, int nrhs, int N);
/* REMORA::uv3dmix (3-D call), TimeIntegration/REMORA_uv3dmix.cpp:6-149 */
int  rh_uv3dmix(rh_ctx *ctx, rh_box xbx, rh_box ybx, rh_array4 u, rh_array4 v, rh_array4 uold, rh_array4 vold,
                rh_array4 rufrc, rh_array4 rvfrc, rh_array4 visc2_p, rh_array4 visc2_r, rh_array4 Hz, rh_array4 pm,
                rh_array4 pn, rh_array4 mskp, int nrhs, int nnew, double dt);
/* REMORA::vert_visc_3d, TimeIntegration/REMORA_vert_visc_3d.cpp:10-107 (no Hzk/AK/BC/DC/FC/CF scratch arguments:
 * the solve lives in registers / thread-private memory) */
int  rh_vert_visc_3d(rh_ctx *ctx, rh_box phi_bx, int ioff, int joff, rh_array4 phi, rh_array4 Hz, rh_array4 Akv,
                     int N, double dt);
/* REMORA::vert_mean_3d, TimeIntegration/REMORA_vert_mean_3d.cpp:10-47 */
int  rh_vert_mean_3d(rh_ctx *ctx, rh_box phi_bx, int ioff, int joff, rh_array4 phi, rh_array4 Hz, rh_array4 Dphi_avg1,
                     rh_array4 dxlen, rh_array4 msk, int N);
/* REMORA::update_massflux_3d, TimeIntegration/REMORA_update_massflux_3d.cpp:24-122 */
int  rh_update_massflux_3d(rh_ctx *ctx, rh_box bx, int ioff, int joff, rh_array4 phi, rh_array4 phibar, rh_array4 Hphi,
                           rh_array4 Hz, rh_array4 pm_or_pn, rh_array4 Dphi_avg1, rh_array4 Dphi_avg2, rh_array4 msk, int N);
/* REMORA::rhs_t_3d, TimeIntegration/REMORA_rhs_t_3d.cpp:27-335 (one tracer) */
int  rh_rhs_t_3d(rh_ctx *ctx, rh_box bx, rh_array4 t, rh_array4 sstore, rh_array4 Huon, rh_array4 Hvom, rh_array4 Hz,
                 rh_array4 pn, rh_array4 pm, rh_array4 W, rh_array4 msku, rh_array4 mskv, int N, double dt);

/* REMORA::rhs_uv_2d, TimeIntegration/REMORA_rhs_uv_2d.cpp:20-238 (rhs_ubar, rhs_vbar -= 2-D momentum advection) */
int  rh_rhs_uv_2d(rh_ctx *ctx, rh_box xbx, rh_box ybx, rh_array4 ubar, rh_array4 vbar, rh_array4 rhs_ubar, rh_array4 rhs_vbar,
                  rh_array4 DUon, rh_array4 DVom, int krhs);

/* ---- drivers (MultiFab level, Source/REMORA.H:357-451) --------------------------------------------- */
/* REMORA::setup_step, TimeIntegration/REMORA_setup_step.cpp:7-369 (+ REMORA::prestep, REMORA_prestep.cpp:36-185) */
int  rh_setup_step(rh_ctx *ctx, rh_state *s, int istep, double t_old);
/* REMORA::advance_2d, TimeIntegration/REMORA_advance_2d.cpp:41-757 (step2d; includes rhs_uv_2d, REMORA_rhs_uv_2d.cpp:20-238) */
int  rh_advance_2d(rh_ctx *ctx, rh_state *s, int istep, double dtfast, int predictor_2d_step, int first_2d_step,
                   int my_iif, int *next_indx1);
/* REMORA::advance_2d_onestep, TimeIntegration/REMORA_advance_2d_onestep.cpp:6-53 */
int  rh_advance_2d_onestep(rh_ctx *ctx, rh_state *s, int istep, int my_iif);
/* REMORA::advance_3d_ml + REMORA::advance_3d, TimeIntegration/REMORA_advance_3d_ml.cpp:6-71, REMORA_advance_3d.cpp:10-387 */
int  rh_advance_3d_ml(rh_ctx *ctx, rh_state *s, int istep);
/* REMORA::Advance, TimeIntegration/REMORA_Advance.cpp:7-35.  The caller swaps old/new first, as REMORA::timeStep does
 * (TimeIntegration/REMORA_TimeStep.cpp:58-61): rh_swap_old_new does that on the views. */
int  rh_advance(rh_ctx *ctx, rh_state *s, int istep, double t_old);
void rh_swap_old_new(rh_state *s);

/* ---- reductions around the step (REMORA::Evolve, Source/REMORA.cpp:139-212) ------------------------------ */
/* ---- GLS turbulence closure (vmix_mode = RH_VMIX_GLS) ---- */
/* REMORA::gls_prestep, TimeIntegration/REMORA_gls.cpp:5-216 (called from setup_step, REMORA_setup_step.cpp:352-357): predictor
 * of tke and gls (component 2) and the Hz-weighted start of the new level (component nnew), then the six ghost fills of
 * :212-215.  W = the omega array of the step (mf_W); nstp = iic % 2, nnew = (iic + 1) % 2. */
int  rh_gls_prestep(rh_ctx *ctx, rh_state *s, rh_array4 W, int nstp, int nnew, int iic, int ntfirst);
/* REMORA::gls_corrector, TimeIntegration/REMORA_gls.cpp:218-915 (called from advance_3d, REMORA_advance_3d.cpp:296-303):
 * advances tke / gls to nnew and rebuilds Akv, Akt, Akk, Akp, Lscale, then the ghost fills of :905-914.  Reads xvel_old /
 * yvel_old, Huon, Hvom, bvf, the surface and bottom stresses.  The per-point pow() calls run on device: results agree with
 * the reference to rounding (tests: rel 1e-12 per step on the turbulence fields), not bit for bit. */
int  rh_gls_corrector(rh_ctx *ctx, rh_state *s, rh_array4 W, int nstp, int nnew);

/* REMORA::estTimeStep, TimeIntegration/REMORA_ComputeTimestep.cpp:39-95: max over the valid cells of |cell-centred velocity| / dx;
 * *dt_out = fixed dt (rh_params.dt) if that is > 0, else cfl / max (1e20 if the max is not positive).  Across the ranks of the
 * built-in RCCL communicator the max is all-reduced (ParallelDescriptor::ReduceRealMax, :66); local_only = 1 returns this rank's
 * cfl / max for hosts that reduce with their own communicator.  Synchronises. */
int  rh_est_timestep(rh_ctx *ctx, rh_state *s, double cfl, int local_only, double *dt_out);
/* REMORA::sum_integrated_quantities + volWgtSumMF, Source/REMORA_SumIQ.cpp:9-141: out4 = { sum scalar*vol, sum kineng*vol, sum vol,
 * max speed } with vol = Hz / (pm pn) and the ROMS kinetic energy of :38-40.  Sums across ranks as above.  Synchronises. */
int  rh_sum_integrated_quantities(rh_ctx *ctx, rh_state *s, int local_only, double out4[4]);
/* The same reduction for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Rank-local (as local_only = 1 above), the four values are written to
 * out4_device (DEVICE memory of the caller, 4 doubles) by kernels enqueued on the context's stream; the call allocates nothing,
 * copies nothing to the host and does not synchronise.  The per-block partials live in a static device array of 41,000 bytes per
 * GPU (fixed launch geometry, so one size serves every box; boxes of 2^32 cells or more are refused with an error).  All calls
 * on one GPU share that array: calls that overlap on one GPU race -- issue them on one stream or order them yourself. */
int  rh_sum_integrated_quantities_noalloc(rh_ctx *ctx, rh_state *s, double *out4_device);

/* ---- synthetic initial state (outside the hot path; what InitData / prob.cpp provide in REMORA) ---- */
/* Fills a bound native state for one of the pinned problems (0 Upwelling, 1 Seamount, 2 DoubleGyre, 3 Advection, 4 DoublyPeriodic, 5 Channel_Test):
 * Source/REMORA.cpp:246-342,608-702, Initialization/REMORA_init.cpp, Exec/<case>/prob.cpp.  Synchronises. */
typedef struct rh_init_params {
    int prob; double rdrag, coriolis_f0, coriolis_beta; int hmix_constant; double visc2, tnu2[RH_NCONS]; double u_0, v_0;
    double z0, zRef, uRef;       /* prob.z0 / prob.zRef / prob.uRef of Exec/DoublyPeriodic (log-layer x-velocity) */
} rh_init_params;
int  rh_init_state(rh_ctx *ctx, rh_state *s, const rh_init_params *ip);
/* plotfile variables of IO/REMORA_Plotfile.cpp:185-235 on the valid box: temp, salt, [scalar], cell-centred u, v, w
 * (device output buffer, FAB order) */
/* REMORA::update_mskp (Initialization/REMORA_make_new_level.cpp:617-655) followed by the periodic ghost fill of mskr / msku / mskv /
 * mskp (REMORA::init_masks_from_netcdf, Initialization/REMORA_init_from_netcdf.cpp:408-412): what the reference does once the
 * land / sea masks have been read.  The caller has written mskr, msku, mskv of the state (1 = water, 0 = land) over their valid
 * boxes and, beyond non-periodic sides of the domain, their ghost cells; ghost cells at rank seams and across periodic sides are
 * filled here (mskr's BEFORE mskp is formed, so that the first valid node line of a box sees its neighbour's cells; the reference
 * gets them from the FAB intersection copy of init_from_netcdf.cpp:400).  rh_params.masks_all_one must be 0 for the step to read them. */
int  rh_update_mskp(rh_ctx *ctx, rh_state *s);
int  rh_plot_vars(rh_ctx *ctx, rh_state *s, int with_scalar, double *out_device);

/* ---- AMReX-native plotfile output (outside the hot path; so that amrex_fcompare / yt / VisIt read the results) ----
 * REMORA::WritePlotFile, WriteMultiLevelPlotfileWithBathymetry, WriteGenericPlotfileHeaderWithBathymetry
 * (Source/IO/REMORA_Plotfile.cpp:141-605), one level, the reference's default variables (temp, salt, [scalar], x/y/z_velocity)
 * plus the nodal mesh-height block "Nu_nd" (z_phys_nd - k dz, REMORA_Plotfile.cpp:320-329). */
typedef struct rh_plot_box {
    rh_box box;                          /* the rank's valid cell box */
    int    file_index;                   /* data went to Level_0/Cell_D_<index> and Level_0/Nu_nd_D_<index> */
    long   cell_offset, nd_offset;       /* byte offsets of the two FAB records in those files */
    double cell_min[6], cell_max[6], nd_min[3], nd_max[3];
} rh_plot_box;
/* This rank's part: gathers the variables on the device, writes one FAB record to each of its two data files under
 * <dir>/Level_0 (created if needed; the files are truncated first, every rank owns the files of its file_index) and fills *out.
 * Synchronises. */
int  rh_plotfile_write_data(rh_ctx *ctx, rh_state *s, const char *dir, int with_scalar, int file_index, rh_plot_box *out);
/* Header, Level_0/Cell_H and Level_0/Nu_nd_H for the boxes of all ranks (host only; needs no device).  A decomposed run gathers
 * the rh_plot_box records with its own communicator and calls this on one rank. */
int  rh_plotfile_write_headers(const rh_params *prm, const char *dir, int with_scalar, int step, double time,
                               int nboxes, const rh_plot_box *boxes);
/* Single-box convenience: both of the above. */
int  rh_write_plotfile(rh_ctx *ctx, rh_state *s, const char *dir, int with_scalar, int step, double time);

/* Checkpoint / restart in the reference's layout (REMORA::WriteCheckpointFile / ReadCheckpointFile, Source/IO/REMORA_Checkpoint.cpp:
 * 15-245, 248-470): text Header + one VisMF (<Name>_H, <Name>_D_00000) per array under <dir>/Level_0, every array with its ghost
 * cells as the reference writes them (Cell, Cell_old, XFace, ..., XRHS, XBar, DU_avg1, zeta, Zt_avg1, h, tke, gls, Lscale, Akk, Akp, Akv,
 * Akt: the reference's full list, :120-202; a state without the GLS closure writes the five closure arrays as init leaves them).  rh_read_checkpoint expects a state that already holds what InitData builds without a
 * checkpoint (rh_init_state or the host's init), overwrites the checkpointed arrays and rebuilds z_w / z_r / Hz; a run continued
 * from it is bit-identical to the uninterrupted one.  One box per call; both synchronise. */
int  rh_write_checkpoint(rh_ctx *ctx, rh_state *s, const char *dir, int step, double time);
int  rh_read_checkpoint(rh_ctx *ctx, rh_state *s, const char *dir, int *step, double *time);

/* per-kernel timing: HIP events recorded on the context's stream around every launch of the named kernel group */
int  rh_timing_enable(rh_ctx *ctx, int on);
int  rh_timing_report(rh_ctx *ctx, char *buf, int buflen);        /* JSON: {"kernel":{"calls":n,"ms":t},...}; synchronises */
int  rh_timing_reset(rh_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
