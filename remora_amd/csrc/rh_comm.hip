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
