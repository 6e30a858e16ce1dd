// rh_diag.hip -- the two reductions that sit around the hot path in REMORA::Evolve (north_star: "wavefront reductions for CFL and
// diagnostics"):
//   rh_est_timestep              : REMORA::estTimeStep, TimeIntegration/REMORA_ComputeTimestep.cpp:39-95 (max-reduce of |u| / dx over
//                                  the cell-centred averages of the face velocities)
//   rh_sum_integrated_quantities : REMORA::sum_integrated_quantities + volWgtSumMF, REMORA_SumIQ.cpp:9-141 (volume-weighted sums of
//                                  the passive scalar, of the ROMS kinetic energy and of 1, and the maximum speed)
// One pass over the valid box serves both: every thread folds its cells into 6 partials, a wavefront folds them with cross-lane
// shuffles, a block through LDS, and the last kernel folds the per-block partials in index order.  No floating-point atomics: the
// result depends on the launch geometry only, and that is fixed (DG_BLOCKS x 256 threads), so it is reproducible run to run.
// Maxima are exact; sums are sums in a different order than the reference's serial MultiFab::Dot (stated tolerance in the tests).
#include "rh_common.h"
#include "rh_rccl.h"

#define DG_BLOCKS 1024
#define DG_THREADS 256
enum { DG_CFL, DG_KEMAX, DG_SCALAR, DG_KINENG, DG_VOLUME, DG_N };

__device__ __forceinline__ double dg_absmax(double a, double b) { return (a < b) ? b : a; }

__global__ void __launch_bounds__(DG_THREADS)
k_diag_partial(B3 bx, A4 cons, A4 u, A4 v, A4 w, A4 Hz, A4 pm, A4 pn, double dxi0, double dxi1, double dxi2, int scalar_comp, double *__restrict__ part)
{
    const long nx = bx.hi0 - bx.lo0 + 1, ny = bx.hi1 - bx.lo1 + 1, nz = bx.hi2 - bx.lo2 + 1, n = nx * ny * nz;
    double cfl = -1.e100, kemax = -1.e300, ssum = 0.0, ksum = 0.0, vsum = 0.0;
    for (long g = (long)blockIdx.x * DG_THREADS + threadIdx.x; g < n; g += (long)DG_BLOCKS * DG_THREADS) {
        const int i = bx.lo0 + (int)(g % nx), j = bx.lo1 + (int)((g / nx) % ny), k = bx.lo2 + (int)(g / (nx * ny));
        const double u0 = u(i, j, k), u1 = u(i + 1, j, k), v0 = v(i, j, k), v1 = v(i, j + 1, k), w0 = w(i, j, k), w1 = w(i, j, k + 1);
        // amrex::average_face_to_cellcenter, then ComputeTimestep.cpp:57-64
        const double cu = 0.5 * (u0 + u1), cv = 0.5 * (v0 + v1), cw = 0.5 * (w0 + w1);
        cfl = dg_absmax(cfl, dg_absmax(dg_absmax(fabs(cu) * dxi0, fabs(cv) * dxi1), fabs(cw) * dxi2));
        const double ke = 0.25 * (u0 * u0 + u1 * u1 + v0 * v0 + v1 * v1);                    // SumIQ.cpp:38-40
        kemax = dg_absmax(kemax, ke);
        const double vol = Hz(i, j, k) / (pm(i, j, 0) * pn(i, j, 0));                         // SumIQ.cpp:131-134
        ssum += cons(i, j, k, scalar_comp) * vol; ksum += ke * vol; vsum += 1.0 * vol;
    }
    // wavefront fold (64 lanes), then the block's 4 wavefronts through LDS
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        cfl = dg_absmax(cfl, __shfl_down(cfl, o)); kemax = dg_absmax(kemax, __shfl_down(kemax, o));
        ssum += __shfl_down(ssum, o); ksum += __shfl_down(ksum, o); vsum += __shfl_down(vsum, o);
    }
    __shared__ double s[DG_THREADS / 64][DG_N];
    const int wv = threadIdx.x / 64;
    if ((threadIdx.x & 63) == 0) { s[wv][DG_CFL] = cfl; s[wv][DG_KEMAX] = kemax; s[wv][DG_SCALAR] = ssum; s[wv][DG_KINENG] = ksum; s[wv][DG_VOLUME] = vsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < DG_THREADS / 64; q++) {
            cfl = dg_absmax(cfl, s[q][DG_CFL]); kemax = dg_absmax(kemax, s[q][DG_KEMAX]);
            ssum += s[q][DG_SCALAR]; ksum += s[q][DG_KINENG]; vsum += s[q][DG_VOLUME];
        }
        double *p = part + (long)blockIdx.x * DG_N;
        p[DG_CFL] = cfl; p[DG_KEMAX] = kemax; p[DG_SCALAR] = ssum; p[DG_KINENG] = ksum; p[DG_VOLUME] = vsum;
    }
}
// per-block partials -> 5 values, blocks folded in index order by one wavefront (lane l owns blocks l, l+64, ...)
__global__ void __launch_bounds__(64) k_diag_final(const double *__restrict__ part, double *__restrict__ out)
{
    double cfl = -1.e100, kemax = -1.e300, ssum = 0.0, ksum = 0.0, vsum = 0.0;
    for (int b = threadIdx.x; b < DG_BLOCKS; b += 64) {
        const double *p = part + (long)b * DG_N;
        cfl = dg_absmax(cfl, p[DG_CFL]); kemax = dg_absmax(kemax, p[DG_KEMAX]); ssum += p[DG_SCALAR]; ksum += p[DG_KINENG]; vsum += p[DG_VOLUME];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        cfl = dg_absmax(cfl, __shfl_down(cfl, o)); kemax = dg_absmax(kemax, __shfl_down(kemax, o));
        ssum += __shfl_down(ssum, o); ksum += __shfl_down(ksum, o); vsum += __shfl_down(vsum, o);
    }
    if (threadIdx.x == 0) { out[DG_CFL] = cfl; out[DG_KEMAX] = kemax; out[DG_SCALAR] = ssum; out[DG_KINENG] = ksum; out[DG_VOLUME] = vsum; }
}

// the two launches of the rank-local reduction of this rank's valid box: per-block partials into `part` (DG_BLOCKS x DG_N doubles,
// every one of them rewritten by every call), the 5 folded values into `out`
static int diag_launch(rh_ctx *c, rh_state *s, double *part, double *out)
{
    const rh_box bx = c->bx;
    RH_NEED(s->cons_new, bx); RH_NEED(s->xvel_new, bx_nodal(bx, 0)); RH_NEED(s->yvel_new, bx_nodal(bx, 1)); RH_NEED(s->zvel_new, bx_nodal(bx, 2));
    RH_NEED(s->Hz, bx); RH_NEED(s->pm, bx_slab(bx, 2, 0)); RH_NEED(s->pn, bx_slab(bx, 2, 0));
    if (s->cons_new.ncomp < RH_NCONS) return rh_fail("diagnostics: cons_new needs %d components", RH_NCONS);
    const rh_params &P = c->prm;
    double dxi[3];
    for (int d = 0; d < 3; d++) dxi[d] = (double)(c->dom.hi[d] - c->dom.lo[d] + 1) / (P.prob_hi[d] - P.prob_lo[d]);   // geom.InvCellSizeArray()
    k_diag_partial<<<DG_BLOCKS, DG_THREADS, 0, c->stream>>>(mkB(bx), mkA(s->cons_new), mkA(s->xvel_new), mkA(s->yvel_new), mkA(s->zvel_new),
        mkA(s->Hz), mkA(s->pm), mkA(s->pn), dxi[0], dxi[1], dxi[2], 2 /* Scalar_comp, REMORA_IndexDefines.H:8-11 */, part);
    RH_LAUNCH_CHECK();
    k_diag_final<<<1, 64, 0, c->stream>>>(part, out);
    RH_LAUNCH_CHECK();
    return 0;
}
// rank-local reduction; host copy of the 5 values (synchronises the context's stream)
static int diag_local(rh_ctx *c, rh_state *s, double h[DG_N])
{
    if (!c->d_diag) RH_HIP(hipMalloc((void **)&c->d_diag, sizeof(double) * (DG_BLOCKS + 2) * DG_N));
    double *part = c->d_diag, *out = c->d_diag + (long)DG_BLOCKS * DG_N;
    RH_TRY(diag_launch(c, s, part, out));
    RH_HIP(hipMemcpyAsync(h, out, sizeof(double) * DG_N, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    return 0;
}
// ParallelDescriptor::ReduceRealMax / ReduceRealSum over the ranks of the built-in RCCL communicator.  With the transport hook the
// host owns the communicator: the caller gets the rank-local values and reduces them itself (as volWgtSumMF(local = true) does).
static int diag_allreduce(rh_ctx *c, double h[DG_N], int local_only)
{
    if (local_only || c->nranks <= 1 || !c->comm) return 0;
    if (!g_rccl.AllReduce) return rh_fail("diagnostics: librccl lacks ncclAllReduce");
    double *buf = c->d_diag + (long)(DG_BLOCKS + 1) * DG_N;
    RH_HIP(hipMemcpyAsync(buf, h, sizeof(double) * DG_N, hipMemcpyHostToDevice, c->stream));
    ncclComm_t comm = (ncclComm_t)c->comm;
    ncclResult_t r = g_rccl.AllReduce(buf, buf, 2, ncclDouble, ncclMax, comm, c->stream);              // DG_CFL, DG_KEMAX
    if (r == ncclSuccess) r = g_rccl.AllReduce(buf + 2, buf + 2, 3, ncclDouble, ncclSum, comm, c->stream);   // the three sums
    if (r != ncclSuccess) return rh_fail("ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
    RH_HIP(hipMemcpyAsync(h, buf, sizeof(double) * DG_N, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rh_est_timestep(rh_ctx *c, rh_state *s, double cfl, int local_only, double *dt_out)
{
    if (!c || !s || !dt_out) return rh_fail("rh_est_timestep: null argument");
    TScope t(c, "est_timestep");
    double h[DG_N];
    RH_TRY(diag_local(c, s, h));
    RH_TRY(diag_allreduce(c, h, local_only));
    double est = 1.e20;                                                                       // ComputeTimestep.cpp:43,67-68
    if (h[DG_CFL] > 0.0) est = cfl / h[DG_CFL];
    *dt_out = (c->prm.dt > 0.0 && !local_only) ? c->prm.dt : est;                             // :90-94 (fixed_dt wins)
    if (local_only) *dt_out = est;
    return 0;
}
extern "C" int rh_sum_integrated_quantities(rh_ctx *c, rh_state *s, int local_only, double out4[4])
{
    if (!c || !s || !out4) return rh_fail("rh_sum_integrated_quantities: null argument");
    TScope t(c, "sum_integrated_quantities");
    double h[DG_N];
    RH_TRY(diag_local(c, s, h));
    RH_TRY(diag_allreduce(c, h, local_only));
    out4[0] = h[DG_SCALAR]; out4[1] = h[DG_KINENG]; out4[2] = h[DG_VOLUME];
    out4[3] = sqrt(2.0 * h[DG_KEMAX]);                                                        // SumIQ.cpp:45
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The entry point that allocates nothing (include/remora_hip.h: rh_sum_integrated_quantities_noalloc; no counterpart in the
// reference).  Same kernels as above; the per-block partials live in a file-scope device array instead of the context's hipMalloc'd
// buffer, and the four results go to a device address of the caller's, so the call issues two kernel launches plus one tiny one and
// nothing else: no hipMalloc / hipFree, no host copy, no synchronisation.
//   scratch: (DG_BLOCKS + 1) x DG_N doubles = 41,000 bytes per GPU, held until the process ends.  The launch geometry is fixed, so
//   the array serves every box size the kernels index (the limit is the 2^32-element bound of RH_NEED, stated in the error).
//   Every GPU holds its own copy at its own address: the address is looked up once per device, with that device current
//   (diag_scratch_init, called by rh_create), and only that pointer -- never the array's name -- is handed to kernels.
//   All calls on one GPU share the array: every call rewrites all of it before reading it (nothing has to start at zero), and two
//   calls that overlap on one GPU (different streams or contexts) race.
#define DG_SCRATCH_DOUBLES ((DG_BLOCKS + 1) * DG_N)
#define DG_NOALLOC_MAX_CELLS (1L << 32)
static __device__ __align__(256) double scratch[DG_SCRATCH_DOUBLES];
#define DG_MAX_DEVICES 64
static double *g_scratch_ptr[DG_MAX_DEVICES];            // per device: the address of `scratch` on that GPU (0 = not looked up yet)
#include <mutex>
static std::mutex g_scratch_mu;

int diag_scratch_init(rh_ctx *c)
{
    int dev = -1;
    RH_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= DG_MAX_DEVICES) return rh_fail("diagnostics: device ordinal %d outside [0, %d)", dev, DG_MAX_DEVICES);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        double *p = nullptr;
        RH_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        if (!p) return rh_fail("diagnostics: hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[dev] = p;
    }
    c->device = dev; c->d_scratch = g_scratch_ptr[dev];
    return 0;
}

// out4 = { sum scalar*vol, sum kineng*vol, sum vol, max speed } on the device (SumIQ.cpp:45 for the last one)
__global__ void k_diag_out4(const double *__restrict__ folded, double *__restrict__ out4)
{
    if (threadIdx.x == 0) { out4[0] = folded[DG_SCALAR]; out4[1] = folded[DG_KINENG]; out4[2] = folded[DG_VOLUME]; out4[3] = sqrt(2.0 * folded[DG_KEMAX]); }
}

extern "C" int rh_sum_integrated_quantities_noalloc(rh_ctx *c, rh_state *s, double *out4_device)
{
    if (!c || !s || !out4_device) return rh_fail("rh_sum_integrated_quantities_noalloc: null argument");
    if (!c->d_scratch) return rh_fail("rh_sum_integrated_quantities_noalloc: the context has no scratch address (rh_create looks it up)");
    int dev = -1;
    RH_HIP(hipGetDevice(&dev));
    if (dev != c->device) return rh_fail("rh_sum_integrated_quantities_noalloc: device %d is current, the context lives on device %d", dev, c->device);
    const long cells = bx_npts(c->bx);
    if (cells >= DG_NOALLOC_MAX_CELLS)
        return rh_fail("rh_sum_integrated_quantities_noalloc: box of %ld cells; this entry point serves boxes below %ld cells (32-bit indexing)", cells, DG_NOALLOC_MAX_CELLS);
    double *part = c->d_scratch, *folded = c->d_scratch + (long)DG_BLOCKS * DG_N;
    RH_TRY(diag_launch(c, s, part, folded));
    k_diag_out4<<<1, 64, 0, c->stream>>>(folded, out4_device);
    RH_LAUNCH_CHECK();
    return 0;
}
