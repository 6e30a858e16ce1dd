// rh_comm.hip -- RCCL communicator of an i-j decomposed run (replaces the MPI side of amrex::FillBoundary).
// RCCL is loaded lazily (dlopen) the first time a communicator is asked for, so the library loads and runs on a
// single GPU without it, and never clashes with another RCCL instance living in the host process.
#include "rh_common.h"
#include <dlfcn.h>
#include "rh_rccl.h"

RcclApi g_rccl;

int rccl_load()
{
    if (g_rccl.handle) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return rh_fail("rh_comm: cannot load librccl.so (%s)", dlerror());
#define RH_SYM(field, sym) do { *(void **)(&g_rccl.field) = dlsym(h, sym); if (!g_rccl.field) { dlclose(h); return rh_fail("rh_comm: librccl lacks %s", sym); } } while (0)
    RH_SYM(GetUniqueId, "ncclGetUniqueId"); RH_SYM(CommInitRank, "ncclCommInitRank"); RH_SYM(CommDestroy, "ncclCommDestroy");
    RH_SYM(GroupStart, "ncclGroupStart"); RH_SYM(GroupEnd, "ncclGroupEnd"); RH_SYM(Send, "ncclSend"); RH_SYM(Recv, "ncclRecv");
    RH_SYM(GetErrorString, "ncclGetErrorString");
#undef RH_SYM
    *(void **)(&g_rccl.AllReduce) = dlsym(h, "ncclAllReduce");          // only the diagnostics of rh_diag.hip use it
    g_rccl.handle = h;
    return 0;
}

extern "C" int rh_comm_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!id128) return rh_fail("rh_comm_unique_id: null buffer");
    RH_TRY(rccl_load());
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rh_fail("ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return 0;
}

extern "C" int rh_comm_init(rh_ctx *c, const void *id128, int rank, int nranks)
{
    if (!c || !id128) return rh_fail("rh_comm_init: null argument");
    if (nranks != c->prm.nranks[0] * c->prm.nranks[1]) return rh_fail("rh_comm_init: nranks %d does not match the %d x %d process grid", nranks, c->prm.nranks[0], c->prm.nranks[1]);
    if (rank != c->prm.rank_xy[1] * c->prm.nranks[0] + c->prm.rank_xy[0]) return rh_fail("rh_comm_init: rank %d is not at (%d,%d) of the process grid", rank, c->prm.rank_xy[0], c->prm.rank_xy[1]);
    RH_TRY(rccl_load());
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    ncclComm_t comm;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return rh_fail("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    c->comm = comm; c->rank = rank; c->nranks = nranks;
    return 0;
}

extern "C" int rh_comm_set_exchange(rh_ctx *c, rh_exchange_fn fn, void *user, int rank, int nranks)
{
    if (!c) return rh_fail("rh_comm_set_exchange: null context");
    if (nranks != c->prm.nranks[0] * c->prm.nranks[1]) return rh_fail("rh_comm_set_exchange: nranks %d does not match the %d x %d process grid", nranks, c->prm.nranks[0], c->prm.nranks[1]);
    if (rank != c->prm.rank_xy[1] * c->prm.nranks[0] + c->prm.rank_xy[0]) return rh_fail("rh_comm_set_exchange: rank %d is not at (%d,%d) of the process grid", rank, c->prm.rank_xy[0], c->prm.rank_xy[1]);
    c->xfn = fn; c->xuser = user; c->rank = rank; c->nranks = nranks;
    return 0;
}

void comm_release(rh_ctx *c)
{
    if (c->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
}

// One ghost exchange: the four message slots of c->d_xbuf, on c->stream.  Sends go out in the order (lower, upper) and
// receives are posted in the order (upper, lower): with two ranks and a periodic direction both neighbours are the same
// peer, and the block sent "down" is the one the peer receives "from above" (messages to one peer match in order).
int comm_exchange(rh_ctx *c, long n_lo, int peer_lo, long n_hi, int peer_hi)
{
    double *s_lo = c->d_xbuf, *s_hi = c->d_xbuf + c->xcap, *r_lo = c->d_xbuf + 2 * c->xcap, *r_hi = c->d_xbuf + 3 * c->xcap;
    // one communicator, two streams (compute and the side stream of halo_async_*): its operations run in issue order on every
    // rank, so an exchange issued on the other stream than the previous one first waits for that one
    if (c->comm_used && c->last_comm_stream != c->stream) RH_HIP(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
    struct Mark { rh_ctx *c; ~Mark() { if (!c->ev_comm) (void)hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming);
                                        (void)hipEventRecord(c->ev_comm, c->stream); c->last_comm_stream = c->stream; c->comm_used = true; } } mark{c};
    if (c->xfn) {
        const int r = c->xfn(c->xuser, s_lo, r_lo, n_lo, n_lo > 0 ? peer_lo : -1, s_hi, r_hi, n_hi, n_hi > 0 ? peer_hi : -1, (void *)c->stream);
        if (r != 0) return rh_fail("ghost exchange: transport hook returned %d", r);
        return 0;
    }
    if (!c->comm) return rh_fail("ghost exchange: no communicator");
    ncclComm_t comm = (ncclComm_t)c->comm;
#define RH_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return rh_fail("%s failed: %s", #x, g_rccl.GetErrorString(r_)); } while (0)
    RH_NCCL(g_rccl.GroupStart());
    if (n_lo > 0) RH_NCCL(g_rccl.Send(s_lo, (size_t)n_lo, ncclDouble, peer_lo, comm, c->stream));
    if (n_hi > 0) RH_NCCL(g_rccl.Send(s_hi, (size_t)n_hi, ncclDouble, peer_hi, comm, c->stream));
    if (n_hi > 0) RH_NCCL(g_rccl.Recv(r_hi, (size_t)n_hi, ncclDouble, peer_hi, comm, c->stream));
    if (n_lo > 0) RH_NCCL(g_rccl.Recv(r_lo, (size_t)n_lo, ncclDouble, peer_lo, comm, c->stream));
    RH_NCCL(g_rccl.GroupEnd());
#undef RH_NCCL
    return 0;
}

// Diagnostic: push two n-double patterns through the active transport to THIS rank (peer_lo = peer_hi = self) and check
// that the block sent "down" arrives "from above" and vice versa -- the message order a two-rank periodic run relies on.
__global__ void k_pattern(double *p, long n, double base) { long q = (long)blockIdx.x * blockDim.x + threadIdx.x; if (q < n) p[q] = base + (double)q; }
extern "C" int rh_comm_selftest(rh_ctx *c, long n)
{
    if (!c || n <= 0) return rh_fail("rh_comm_selftest: bad argument");
    if (!c->xfn && !c->comm) return rh_fail("rh_comm_selftest: no communicator");
    if (n > c->xcap) {
        RH_HIP(hipStreamSynchronize(c->stream));
        if (c->d_xbuf) RH_HIP(hipFree(c->d_xbuf));
        c->xcap = (n + 4095) / 4096 * 4096; c->d_xbuf = nullptr;
        RH_HIP(hipMalloc((void **)&c->d_xbuf, sizeof(double) * 4 * c->xcap));
    }
    RH_HIP(hipMemsetAsync(c->d_xbuf, 0, sizeof(double) * 4 * c->xcap, c->stream));
    const int nb = (int)((n + 255) / 256);
    k_pattern<<<nb, 256, 0, c->stream>>>(c->d_xbuf, n, 1.0e6);                 // send_lo
    k_pattern<<<nb, 256, 0, c->stream>>>(c->d_xbuf + c->xcap, n, 2.0e6);       // send_hi
    RH_LAUNCH_CHECK();
    RH_TRY(comm_exchange(c, n, c->rank, n, c->rank));
    std::vector<double> lo(n), hi(n);
    RH_HIP(hipMemcpyAsync(lo.data(), c->d_xbuf + 2 * c->xcap, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipMemcpyAsync(hi.data(), c->d_xbuf + 3 * c->xcap, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    RH_HIP(hipStreamSynchronize(c->stream));
    for (long q = 0; q < n; q++) {
        if (hi[q] != 1.0e6 + (double)q) return rh_fail("rh_comm_selftest: recv_hi[%ld] = %g, expected the block sent down (%g)", q, hi[q], 1.0e6 + (double)q);
        if (lo[q] != 2.0e6 + (double)q) return rh_fail("rh_comm_selftest: recv_lo[%ld] = %g, expected the block sent up (%g)", q, lo[q], 2.0e6 + (double)q);
    }
    return 0;
}
