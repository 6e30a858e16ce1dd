// rh_common.h -- shared device/host helpers of the MI355X REMORA hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include "../../include/remora_hip.h"

// Device-side view with the amrex::Array4 memory order (i fastest).  Built once per launch on the host.
// Indexing is 32-bit unsigned: the offset then rides in one VGPR next to a scalar base pointer (global_load ... v, s[base]),
// which halves the address arithmetic of these load-heavy stencils.  Every array that reaches a kernel passes RH_NEED first,
// which rejects arrays of 2^32 elements or more (34 GB; a state of ~35 such components would not fit the 288 GB of one GPU).
struct A4 {
    double *p; int lo0, lo1, lo2, hi0, hi1, hi2, nc; unsigned sy, sz, sn;
    __host__ __device__ __forceinline__ unsigned ix(int i, int j, int k) const
    { return (unsigned)(i - lo0) + (unsigned)(j - lo1) * sy + (unsigned)(k - lo2) * sz; }
    __device__ __forceinline__ double &operator()(int i, int j, int k) const { return p[ix(i, j, k)]; }
    __device__ __forceinline__ double &operator()(int i, int j, int k, int n) const { return p[ix(i, j, k) + (unsigned)n * sn]; }
};

static inline A4 mkA(const rh_array4 &a)
{
    A4 r; r.p = a.p; r.lo0 = a.lo[0]; r.lo1 = a.lo[1]; r.lo2 = a.lo[2]; r.hi0 = a.hi[0]; r.hi1 = a.hi[1]; r.hi2 = a.hi[2];
    r.nc = a.ncomp; r.sy = (unsigned)(a.hi[0] - a.lo[0] + 1); r.sz = r.sy * (unsigned)(a.hi[1] - a.lo[1] + 1); r.sn = r.sz * (unsigned)(a.hi[2] - a.lo[2] + 1);
    return r;
}
static inline A4 mkA(const rh_array4 &a, int comp) { A4 r = mkA(a); r.p += (long)r.sn * comp; r.nc -= comp; return r; }
static inline long a4_size(const rh_array4 &a)
{ return (long)(a.hi[0]-a.lo[0]+1)*(a.hi[1]-a.lo[1]+1)*(a.hi[2]-a.lo[2]+1)*a.ncomp; }
static inline rh_array4 a4_comp(rh_array4 a, int comp)
{ a.p += (long)(a.hi[0]-a.lo[0]+1)*(a.hi[1]-a.lo[1]+1)*(a.hi[2]-a.lo[2]+1)*comp; a.ncomp -= comp; return a; }

struct B3 { int lo0, lo1, lo2, hi0, hi1, hi2; };
static inline B3 mkB(const rh_box &b) { return B3{b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2]}; }
static inline rh_box bx_make(int l0,int l1,int l2,int h0,int h1,int h2){ rh_box b={{l0,l1,l2},{h0,h1,h2}}; return b; }
static inline rh_box bx_grow(rh_box b,int g0,int g1,int g2){ b.lo[0]-=g0;b.hi[0]+=g0;b.lo[1]-=g1;b.hi[1]+=g1;b.lo[2]-=g2;b.hi[2]+=g2; return b; }
static inline rh_box bx_nodal(rh_box b,int d){ b.hi[d]+=1; return b; }
static inline rh_box bx_slab(rh_box b,int d,int v){ b.lo[d]=v; b.hi[d]=v; return b; }
static inline rh_box bx_and(rh_box a, rh_box b){ for(int d=0;d<3;d++){ if(b.lo[d]>a.lo[d])a.lo[d]=b.lo[d]; if(b.hi[d]<a.hi[d])a.hi[d]=b.hi[d]; } return a; }
static inline rh_box bx_union(rh_box a, rh_box b){ for(int d=0;d<3;d++){ if(b.lo[d]<a.lo[d])a.lo[d]=b.lo[d]; if(b.hi[d]>a.hi[d])a.hi[d]=b.hi[d]; } return a; }
static inline bool  bx_ok(rh_box b){ return b.lo[0]<=b.hi[0]&&b.lo[1]<=b.hi[1]&&b.lo[2]<=b.hi[2]; }
static inline rh_box bx_of(const rh_array4 &a){ rh_box b; for(int d=0;d<3;d++){b.lo[d]=a.lo[d];b.hi[d]=a.hi[d];} return b; }
static inline bool bx_contains(rh_box a, rh_box b){ for(int d=0;d<3;d++) if(b.lo[d]<a.lo[d]||b.hi[d]>a.hi[d]) return false; return true; }
static inline rh_box bx_growlo_i(rh_box b){ b.lo[0]-=1; return b; }
static inline rh_box bx_growlo_j(rh_box b){ b.lo[1]-=1; return b; }
static inline long bx_npts(rh_box b){ return bx_ok(b) ? (long)(b.hi[0]-b.lo[0]+1)*(b.hi[1]-b.lo[1]+1)*(b.hi[2]-b.lo[2]+1) : 0; }

__device__ __forceinline__ bool inb(const B3 &b, int i, int j, int k)
{ return i >= b.lo0 && i <= b.hi0 && j >= b.lo1 && j <= b.hi1 && k >= b.lo2 && k <= b.hi2; }
__device__ __forceinline__ bool inb2(const B3 &b, int i, int j)
{ return i >= b.lo0 && i <= b.hi0 && j >= b.lo1 && j <= b.hi1; }

// std::max / std::min semantics of the reference (first argument on ties)
__device__ __forceinline__ double dmax_(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double dmin_(double a, double b) { return (b < a) ? b : a; }

// Launch geometry.  x threads run along i (coalesced 512-B rows per wave), RH_BY rows of j per block.
#define RH_BX 64
#define RH_BY 4
static inline dim3 rh_block() { return dim3(RH_BX, RH_BY, 1); }
// pointwise 3-D kernels: grid.z walks k
static inline dim3 grid3(const rh_box &b)
{ return dim3((b.hi[0]-b.lo[0]+RH_BX)/RH_BX, (b.hi[1]-b.lo[1]+RH_BY)/RH_BY, b.hi[2]-b.lo[2]+1); }
// column kernels / 2-D kernels: one thread per (i,j)
static inline dim3 grid2(const rh_box &b)
{ return dim3((b.hi[0]-b.lo[0]+RH_BX)/RH_BX, (b.hi[1]-b.lo[1]+RH_BY)/RH_BY, 1); }
// XCD-aware tile order (cdna_hip_programming.md T1).  Workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest),
// each XCD with a private L2, so tiles that are neighbours in memory land on different L2s and every shared row or halo is
// fetched over the fabric once per sharer (measured: 1.4-3x the algorithmic bytes, profiles/r1_v3_traffic.json).  This
// bijective remap hands each group of workgroups that share an XCD a contiguous run of tiles (x fastest, then y, then z), so the
// neighbours' rows are found in the XCD's own L2.  Pure performance choice: any mapping gives the same results.
struct RhTile { int x, y, z; };
__device__ __forceinline__ RhTile rh_tile()
{
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    const unsigned orig = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
    const unsigned q = n / 8, r = n % 8, xcd = orig % 8;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    RhTile o; o.x = (int)(t % gx); const unsigned u = t / gx; o.y = (int)(u % gy); o.z = (int)(u / gy);
    return o;
}
// Variant for the k-marching tiled kernels.  Their blocks live for the whole column march, and two tiles share halo rows level by
// level, so sharing an L2 only pays if both are at the same level at the same time.  The blocks resident on an XCD start together
// (and, doing equal work, retire together), so the XCD's run of tiles is walked in compact PX x PY patches: a patch is what is
// resident at once, and most halo rows are shared inside it.
template <int PX, int PY>
__device__ __forceinline__ RhTile rh_tile_patch()
{
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy;
    const unsigned orig = blockIdx.y * gx + blockIdx.x;
    const unsigned q = n / 8, r = n % 8, xcd = orig % 8;
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    const unsigned pr = t / (gx * PY), rem = t - pr * gx * PY;                  // patch row (every earlier one is full height)
    const unsigned h = min((unsigned)PY, gy - pr * PY);
    const unsigned pc = rem / (PX * h), rem2 = rem - pc * PX * h;               // patch in that row (every earlier one is full width)
    const unsigned w = min((unsigned)PX, gx - pc * PX);
    const unsigned ly = rem2 / w, lx = rem2 - ly * w;
    RhTile o; o.x = (int)(pc * PX + lx); o.y = (int)(pr * PY + ly); o.z = 0;
    return o;
}
#define RH_IJ(b) const RhTile rh_t_ = rh_tile(); const int i = (b).lo0 + rh_t_.x * RH_BX + (int)threadIdx.x; const int j = (b).lo1 + rh_t_.y * RH_BY + (int)threadIdx.y
#define RH_KZ rh_t_.z      /* grid.z index of a pointwise 3-D kernel (after RH_IJ) */

// Dynamic LDS of the one-wavefront column kernels.  Up to 64 KB needs no opt-in; above that (deep columns: nz >= 63 for the
// tridiagonal solver) the kernel is raised to what it asks for, up to half the CU's 160 KB so that two wavefronts still share a CU.
#define RH_LDS_MAX (80 * 1024)
template <class K> static inline hipError_t rh_lds_optin(K kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
extern thread_local std::string g_rh_err;
int rh_fail(const char *fmt, ...);
#define RH_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return rh_fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define RH_TRY(x) do { int r_ = (x); if (r_) return r_; } while (0)
#define RH_LAUNCH_CHECK() RH_HIP(hipGetLastError())
// operand-shape guard used before every launch: the box a kernel indexes must lie inside the array it indexes
#define RH_NEED(arr, box) do { if (a4_size(arr) >= (1L << 32)) return rh_fail("%s: array %s has %ld elements; 32-bit indexing needs < 2^32 (decompose further)", __func__, #arr, a4_size(arr)); \
    if (!bx_contains(bx_of(arr), (box))) return rh_fail("%s: array %s [%d:%d,%d:%d,%d:%d] does not cover box [%d:%d,%d:%d,%d:%d] (%s:%d)", __func__, #arr, \
    (arr).lo[0],(arr).hi[0],(arr).lo[1],(arr).hi[1],(arr).lo[2],(arr).hi[2],(box).lo[0],(box).hi[0],(box).lo[1],(box).hi[1],(box).lo[2],(box).hi[2], __FILE__, __LINE__); } while (0)

struct BCRec3 { int lo[3], hi[3]; };
void *bc_queue_new();
void bc_queue_delete(void *q);
void *halo_queue_new();
void halo_queue_delete(void *q);

// a view plus what AMReX keeps in the MultiFab: valid box, index type, ghost widths
struct MFV { rh_array4 a; rh_box valid; int ix[3]; int ng[3]; };

struct rh_ctx {
    rh_params prm;
    hipStream_t stream;
    int N, nz;
    rh_box dom, bx;                  // domain and this rank's valid cell box
    int nfast; std::vector<double> w1, w2;
    BCRec3 bcs[RH_BCV_NUM]; double extdir[RH_BCV_NUM][6];
    int phys[10][6]; double physval[10][6];      // phys_bc_type per user variable and face (init_bcs.cpp:9-199), inflow values
    // stretching curves (host-computed, device copy): [4][nz+1] = sc_w, Cs_w, sc_r, Cs_r
    std::vector<double> h_sc; double *d_sc = nullptr;
    // device work space (replaces the per-call MultiFab / FArrayBox temporaries of the reference)
    rh_array4 rho, P, W, wCF, wDC, DUon, DVom, Drhs, Dnew, zwrk, gzeta, gzeta2, gzetaSA;
    rh_array4 gW1, gW2, gW3, gS2C, gB2, gS2;      // GLS corrector scratch (allocated only with vmix_mode = RH_VMIX_GLS)
    double *d_work = nullptr; long work_bytes = 0;
    // RCCL
    void *comm = nullptr; int rank = 0, nranks = 1;
    rh_exchange_fn xfn = nullptr; void *xuser = nullptr;      // transport of the ghost exchange (RCCL by default)
    double *d_diag = nullptr;                                  // per-block partials of the diagnostics reductions (rh_diag.hip)
    int device = -1; double *d_scratch = nullptr;              // this context's GPU and the address of rh_diag.hip's static scratch array on it
    double *d_xbuf = nullptr; long xcap = 0;                   // 4 message slots: send_lo, send_hi, recv_lo, recv_hi (xcap doubles each)
    // side stream for ghost fills whose consumer is far away (rh_halo.hip: halo_async_*): the fill scope forks off the compute
    // stream, runs pack -> send/recv -> unpack -> BC passes on xstream with its own message slots, and is joined before the consumer
    hipStream_t own_xstream = nullptr;                         // the side stream this context created (never swapped; rh_destroy frees this one only)
    hipStream_t xstream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_comm = nullptr;
    double *d_xbuf2 = nullptr; long xcap2 = 0;
    // barotropic loop of a decomposed run (rh_step2d.hip): the new level's ghost fill of call n is owed until call n + 1 has put its
    // interior tiles on the compute stream; it then runs on the side stream, under them
    struct { bool on = false; rh_array4 ubar, vbar, zeta, msku, mskv, mskr; int knew = 0; } pend2d;
    hipEvent_t ev2d_m = nullptr, ev2d_b = nullptr;
    bool no_mffuse = false;                                    // RH_NO_MFFUSE=1: update_massflux_3d always as its own launches
    bool in_loop2d = false, no_split2d = false;                // RH_NO_SPLIT2D=1: never split k2d_fused into interior and rim launches
    bool async_on = false, async_pending = false, in_advance = false, no_async = false, async_rccl = false;
    hipStream_t last_comm_stream = nullptr; bool comm_used = false;
    // per-kernel-group timing
    bool timing = false;
    std::vector<std::string> tnames; std::map<std::string,int> tslot;
    std::vector<std::vector<std::pair<hipEvent_t,hipEvent_t>>> tev;
    std::vector<hipEvent_t> evpool;
    void *bcq = nullptr; int bc_defer = 0;                     // queued physical-BC passes (rh_halo.hip)
    void *haloq = nullptr;                                     // queued ghost exchanges of the same scope (rh_halo.hip)
    bool P_ready = false;            // prsgrd's hydrostatic pressure was already formed by launch_rho_eos_P (rh_setup.hip)
    bool noregsolve = false;         // RH_NOREGSOLVE=1: the LDS-ring column solver instead of the register form (A/B parity and timing)
    bool nofuse2d = false;           // RH_NOFUSE2D=1: advance_2d as three kernels + the DUon/DVom exchange instead of k2d_fused (A/B parity)
    bool unfused = false;            // RH_UNFUSED=1: drivers call the per-leaf entry points instead of the fused kernels (A/B parity)
    bool gls_libm_pow = false;       // RH_GLS_LIBM_POW=1: every per-point power of the GLS closure through pow() (A/B parity and timing)
    bool no_bc2d = false;            // RH_NO_BC2D=1: the ghost cells of the barotropic step always through the batched BC passes, never in k2d_fused (A/B parity and timing)
    bool no_rhofuse = false;         // RH_NO_RHOFUSE=1: rho_eos (+ P) and prsgrd as two launches handing rho and P over through memory (A/B parity and timing)
    bool poison = false;             // RH_POISON=1: NaN-fill the work space at the start of every step (stale-read detector)
    // RH_VMIX_CONST (init_custom_vmix of the Seamount / DoubleGyre kind): set_analytical_vmix (REMORA.cpp:565-573) rewrites the same
    // constants into Akv / Akt at every step.  Here k_vmix and its FillPatch run once per bound state (vmix_akv_done / vmix_akt_done
    // remember which arrays hold the constants), and the column solvers the drivers launch read the constants from four short rows
    // (d_vmixrow: Akv, Akt x 3; zero row / level strides, so every load hits L1) instead of streaming the 3-D arrays.  Same values,
    // same expressions: bit-identical.  RH_VMIX_REWRITE=1 goes back to the per-step rewrite and the array reads (A/B parity and timing).
    double *d_vmixrow = nullptr; int vmixrow_len = 0; const double *vmix_akv_done = nullptr, *vmix_akt_done = nullptr; bool vmix_rewrite = false; bool no_ckonst = false, no_flatstretch = false, no_xshare = false, no_k2mode = false;   // RH_NO_CKONST=1: the register solvers read the constant rows instead of taking the constant as a number (A/B parity)
   
    ~rh_ctx() { if (bcq) bc_queue_delete(bcq); if (haloq) halo_queue_delete(haloq); }
};

// view of Akv (which = 0) or of component n of Akt (which = 1 + n) for the column solvers the drivers launch: the constant rows
// when the arrays are known to hold the constants (see rh_ctx::d_vmixrow), the arrays themselves otherwise
// the number a constant-row view of mkA_vmix stands for (false: `a` is a real array)
static inline bool vmix_row_value(const rh_ctx *c, const A4 &a, double *v)
{
    if (!c->d_vmixrow || c->vmixrow_len <= 0 || a.sz != 0 || a.sy != 0) return false;
    const long off = a.p - c->d_vmixrow;
    if (off < 0 || off >= 4L * c->vmixrow_len) return false;
    const int which = (int)(off / c->vmixrow_len);
    *v = which == 0 ? c->prm.Akv_const : c->prm.Akt_const[which - 1];
    return true;
}
static inline A4 mkA_vmix(const rh_ctx *c, const rh_state *s, int which)
{
    const rh_array4 &real = which == 0 ? s->Akv : s->Akt;
    A4 r = which == 0 ? mkA(real) : mkA(real, which - 1);
    const bool rows = c->prm.vmix_mode == RH_VMIX_CONST && !c->vmix_rewrite && c->d_vmixrow && c->vmix_akv_done == s->Akv.p &&
                      c->vmix_akt_done == s->Akt.p && real.hi[0] - real.lo[0] + 1 <= c->vmixrow_len;
    if (rows) { r.p = c->d_vmixrow + (long)which * c->vmixrow_len; r.sy = 0; r.sz = 0; r.sn = 0; }
    return r;
}
int  vert_visc_3d_view(rh_ctx *c, rh_box phi_bx, int ioff, int joff, rh_array4 phi, rh_array4 Hz, rh_array4 Akv, A4 akv_view, int N, double dt);   // rh_adv3d.hip

// timing scope: records events around a kernel group when enabled
struct TScope {
    rh_ctx *c; hipEvent_t a, b; int slot; bool on;
    TScope(rh_ctx *ctx, const char *name);
    ~TScope();
};

// ---- internal entry points shared between the .hip files ----
MFV  mfv(const rh_ctx *c, const rh_array4 &a, int ix0, int ix1, int ix2);
// halo / bc (rh_halo.hip)
int  halo_fill_boundary(rh_ctx *c, const MFV &m, int icomp, int ncomp);
// calc / msku / mskv: what the radiation conditions read (PhysBCFunct.cpp:21-23); null = the reference's empty MultiFab
int  halo_physbc(rh_ctx *c, const MFV &m, const rh_array4 &msk, int icomp, int ncomp, int bccomp, int n_not_fill,
                 const rh_array4 *calc = nullptr, const rh_array4 *msku = nullptr, const rh_array4 *mskv = nullptr);
int  halo_fillpatch(rh_ctx *c, const MFV &m, const rh_array4 &msk, int bccomp, int icomp, int ncomp,
                    const rh_array4 *calc = nullptr, const rh_array4 *msku = nullptr, const rh_array4 *mskv = nullptr);
int  halo_flush(rh_ctx *c);
// fork: later ghost fills run on the side stream (after everything queued on the compute stream so far); end: back to the compute
// stream, which does NOT wait; join: the compute stream waits for the side stream's fills (no-op when nothing is pending)
int  halo_async_begin(rh_ctx *c);
int  halo_async_end(rh_ctx *c);
int  halo_async_join(rh_ctx *c);
int  halo_side_enter(rh_ctx *c);     // the following launches, messages and BC passes go to the side stream ...
int  halo_side_leave(rh_ctx *c);     // ... until here (no events: the caller orders the two streams)
int  step2d_flush_pending(rh_ctx *c);   // the owed ghost fill of the barotropic loop, on the current stream
// Ghost exchanges and BC passes issued between bc_begin and the matching bc_end are queued and flushed as one launch per dependency level; only
// ghost-fill calls may sit inside such a scope (any other kernel on the same arrays would overtake the queued passes)
void bc_begin(rh_ctx *c);
int  bc_end(rh_ctx *c);
int  bc_flush(rh_ctx *c);
// side-stream scope of halo_async_begin / halo_async_end: the destructor swaps the context back on every exit path
struct AsyncScope { rh_ctx *c; AsyncScope(rh_ctx *ctx) : c(ctx) {} ~AsyncScope() { if (c) (void)halo_async_end(c); } int begin() { return halo_async_begin(c); } int end() { rh_ctx *x = c; c = nullptr; return halo_async_end(x); } };
struct BcScope { rh_ctx *c; BcScope(rh_ctx *ctx) : c(ctx) { bc_begin(c); } ~BcScope() { if (c) (void)bc_end(c); } int end() { rh_ctx *x = c; c = nullptr; return bc_end(x); } };
// kernels without a C entry point of their own (rh_setup.hip, rh_adv3d.hip, rh_step2d.hip)
int  launch_bstress(rh_ctx *c, rh_state *s);
int  launch_massflux_old(rh_ctx *c, rh_state *s);
int  launch_massflux_omega(rh_ctx *c, rh_state *s, rh_box gbx1, int N, int *applied);
int  launch_vmix(rh_ctx *c, rh_state *s);
int  launch_set_zeta(rh_ctx *c, rh_state *s);
int  launch_uv_ab3(rh_ctx *c, rh_state *s, int istep);
int  launch_uv_corrector(rh_ctx *c, rh_state *s, int istep, int *applied);   // fused: AB3 + vert_visc_3d + vert_mean_3d for u, v (*applied = 2: + update_massflux_3d on the valid non-wall faces)
int  launch_massflux_rest(rh_ctx *c, rh_box full, rh_box done, int ioff, int joff, rh_array4 phi, rh_array4 phibar, rh_array4 Hphi,
                          rh_array4 Hz, rh_array4 pm_or_pn, rh_array4 Dphi_avg1, rh_array4 Dphi_avg2, rh_array4 msk, int N);
int  launch_mask_cons(rh_ctx *c, rh_state *s);
int  launch_rho_eos_P(rh_ctx *c, rh_state *s, rh_box bx, rh_box pbx);
int  launch_rho_prsgrd(rh_ctx *c, rh_state *s, rh_box gbx2, rh_box xbx, rh_box ybx, int nrhs, int N, int *applied);
int  launch_prestep_uv2(rh_ctx *c, rh_state *s, rh_box bx, rh_box dead_u, rh_box dead_v, int iic, int ntfirst, int nrhs, int N, double dt);
int  launch_prestep_uv(rh_ctx *c, rh_box vel_bx, rh_box dead, int ioff, int joff, rh_array4 vel, rh_array4 vel_old, rh_array4 rvel,
                       rh_array4 Hz, rh_array4 sstr, rh_array4 bstr, rh_array4 pm, rh_array4 pn, int iic, int ntfirst, int nrhs, int N, double dt);
int  launch_prestep_t3(rh_ctx *c, rh_state *s, int iic, int ntfirst);   // fused: 3 tracers x (prestep_t + t3dmix)
int  launch_rhs_t3(rh_ctx *c, rh_state *s);                             // fused: 3 tracers x rhs_t_3d
void comm_release(rh_ctx *c);                                   // rh_comm.hip
int  comm_exchange(rh_ctx *c, long n_lo, int peer_lo, long n_hi, int peer_hi);   // d_xbuf slots, on c->stream
// small utilities (rh_ctx.hip)
int  dev_setval(rh_ctx *c, const rh_array4 &a, int icomp, int ncomp, double v);
int  dev_copy(rh_ctx *c, const rh_array4 &dst, const rh_array4 &src);
int  dev_setbox(rh_ctx *c, const rh_array4 &a, int comp, rh_box b, double v);      // rh_driver.hip
int  diag_scratch_init(rh_ctx *c);                                                 // rh_diag.hip
int  launch_gls_init(rh_ctx *c, rh_state *s);                                      // rh_gls.hip
