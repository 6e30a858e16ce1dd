// rh_halo.hip -- ghost-cell fills on device.
//   halo_fill_boundary : amrex::FabArray::FillBoundary(periodicity) for a rank that owns the whole periodic extent
//                        (what FillPatchSingleLevel reduces to at level 0, BoundaryConditions/REMORA_FillPatch.cpp:96-102)
//   halo_physbc        : REMORAPhysBCFunct::operator() (BoundaryConditions/REMORA_PhysBCFunct.cpp:21-93) with
//                        impose_{cons,xvel,yvel,zvel}_bcs (REMORA_BoundaryConditions_{cons,xvel,yvel,zvel}.cpp)
//                        for int_dir / foextrap / ext_dir sides (periodic + SlipWall)
#include "rh_common.h"

// ---- periodic wrap: one thread per ghost point of the frame around the valid box -------------------------------
// Ghost points are enumerated row-wise: first the full-width rows below/above the valid rows, then the two side
// segments of every valid row.  Source search order (sx, then sy, -1..1) follows the single-box FillBoundary.
// up to P_BATCH arrays of one fill scope per launch (the three fills that end a barotropic call are one launch, not three)
#define P_BATCH 8
struct PFill { A4 a; B3 v; int icomp, ncomp; };
struct PBatch { int n, perx, pery, lenx, leny; PFill q[P_BATCH]; };
__global__ void k_fill_periodic(PBatch pb)
{
    const A4 &a = pb.q[blockIdx.y].a; const B3 &v = pb.q[blockIdx.y].v;
    const int icomp = pb.q[blockIdx.y].icomp, ncomp = pb.q[blockIdx.y].ncomp;
    const int perx = pb.perx, pery = pb.pery, lenx = pb.lenx, leny = pb.leny;
    const int NX = a.hi0 - a.lo0 + 1, NY = a.hi1 - a.lo1 + 1;
    const int vx = v.hi0 - v.lo0 + 1, vy = v.hi1 - v.lo1 + 1;
    const int nglo_x = v.lo0 - a.lo0, nglo_y = v.lo1 - a.lo1;
    const long nrow_part = (long)(NY - vy) * NX;
    const long per_k = nrow_part + (long)vy * (NX - vx);
    const int nk = a.hi2 - a.lo2 + 1;
    const long total = per_k * nk * ncomp;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const int n = icomp + (int)(g / (per_k * nk));
        long r = g % (per_k * nk);
        const int k = a.lo2 + (int)(r / per_k);
        r = r % per_k;
        int i, j;
        if (r < nrow_part) {
            const int row = (int)(r / NX); i = a.lo0 + (int)(r % NX);
            j = row < nglo_y ? a.lo1 + row : v.hi1 + 1 + (row - nglo_y);
        } else {
            r -= nrow_part;
            const int w = NX - vx; const int row = (int)(r / w); const int cc = (int)(r % w);
            j = v.lo1 + row;
            i = cc < nglo_x ? a.lo0 + cc : v.hi0 + 1 + (cc - nglo_x);
        }
        bool done = false;
        for (int sx = -1; sx <= 1 && !done; sx++) {
            if (sx != 0 && !perx) continue;
            for (int sy = -1; sy <= 1 && !done; sy++) {
                if (sy != 0 && !pery) continue;
                if (sx == 0 && sy == 0) continue;
                const int ii = i + sx * lenx, jj = j + sy * leny;
                if (ii >= v.lo0 && ii <= v.hi0 && jj >= v.lo1 && jj <= v.hi1) { a(i, j, k, n) = a(ii, jj, k, n); done = true; }
            }
        }
    }
}

// ---- i-j decomposition: ghost blocks exchanged with the neighbours of the process grid -------------------------------
// FillBoundary across ranks runs in two phases, X then Y.  Phase X swaps ghost COLUMNS with the left /
// right neighbour over the valid rows only; phase Y swaps ghost ROWS with the lower / upper neighbour over the full width,
// ghost columns included -- so a corner block reaches its diagonal owner in two hops and no diagonal message exists.  A
// direction owned by a single rank wraps locally (k_fill_periodic / k_wrap_rows).  Geometry only (host, no device).  With an
// index type that is nodal in the exchanged direction the last valid face (hi + 1) is shared with the upper neighbour and is
// not exchanged: FillBoundary fills ghost points only.
static int plan_dir(const rh_params &P, int d, const int alo[3], const int ahi[3], const int ng[3], rh_halo_plan *o)
{
    const int o_ = 1 - d;                                                       // the other horizontal direction
    const int np = P.nranks[d] < 1 ? 1 : P.nranks[d], r = P.rank_xy[d];
    const int npo = P.nranks[o_] < 1 ? 1 : P.nranks[o_], ro = P.rank_xy[o_];
    const int vlo = alo[d] + ng[d], vhi = ahi[d] - ng[d];                       // valid range of this array along d
    const int id = (vhi - vlo) - (P.box_hi[d] - P.box_lo[d]);                   // 1 if nodal along d
    if (id != 0 && id != 1) return rh_fail("halo plan: array range [%d,%d] with ghost width %d does not match the rank's box range [%d,%d] (direction %d)", alo[d], ahi[d], ng[d], P.box_lo[d], P.box_hi[d], d);
    if (vlo != P.box_lo[d]) return rh_fail("halo plan: array does not start at the rank's first %s", d ? "row" : "column");
    o->nrows = ng[d];
    int c_lo = (r > 0) ? r - 1 : (P.periodic[d] ? np - 1 : -1);
    int c_hi = (r < np - 1) ? r + 1 : (P.periodic[d] ? 0 : -1);
    if (np == 1) c_lo = c_hi = -1;                                              // a single rank wraps locally
    auto rank_of = [&](int cd) { return cd < 0 ? -1 : (d == 1 ? cd * npo + ro : ro * np + cd); };   // rank = ry * npx + rx
    o->peer_lo = rank_of(c_lo); o->peer_hi = rank_of(c_hi);
    const int volo = alo[o_] + ng[o_], vohi = ahi[o_] - ng[o_];
    if (d == 0) { o->i_lo = volo; o->i_hi = vohi; }                             // phase X: valid rows only
    else {                                                                      // phase Y: ghost columns travel where FillBoundary fills them
        const bool left = P.periodic[0] || P.rank_xy[0] > 0, right = P.periodic[0] || P.rank_xy[0] < (P.nranks[0] < 1 ? 1 : P.nranks[0]) - 1;
        o->i_lo = left ? alo[0] : volo; o->i_hi = right ? ahi[0] : vohi;
    }
    o->recv_lo_j = vlo - ng[d];            o->recv_hi_j = vhi + 1;
    o->send_hi_j = P.box_hi[d] + 1 - ng[d];                                     // the upper neighbour's lower ghost block
    o->send_lo_j = P.box_lo[d] + id;                                            // the lower neighbour's upper ghost block
    if ((o->peer_lo >= 0 || o->peer_hi >= 0) && ng[d] > P.box_hi[d] - P.box_lo[d] + 1 - id) return rh_fail("halo plan: box thinner than the ghost width (direction %d)", d);
    return 0;
}
extern "C" int rh_halo_plan_j(const rh_params *prm, const int arr_lo[3], const int arr_hi[3], const int ng[3], rh_halo_plan *out)
{
    if (!prm || !out) return rh_fail("rh_halo_plan_j: null argument");
    return plan_dir(*prm, 1, arr_lo, arr_hi, ng, out);
}
extern "C" int rh_halo_plan_i(const rh_params *prm, const int arr_lo[3], const int arr_hi[3], const int ng[3], rh_halo_plan *out)
{
    if (!prm || !out) return rh_fail("rh_halo_plan_i: null argument");
    return plan_dir(*prm, 0, arr_lo, arr_hi, ng, out);
}

// One block of one array travelling through a message slot: [i0, i0+w) x [j0, j0+h) x all k x ncomp comps, packed i fastest.
struct XDesc { double *p; double *buf; unsigned sy, sz, sn; int i0, j0, w, h, nk, ncomp; };
#define X_BATCH 16
struct XBatch { int n, unpack; XDesc q[X_BATCH]; };
__global__ void k_xblocks(XBatch b)
{
    const XDesc &q = b.q[blockIdx.y];
    const long w = q.w, h = q.h, total = w * h * q.nk * q.ncomp;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const unsigned i = (unsigned)(g % w); long r = g / w;
        const unsigned j = (unsigned)(r % h); r /= h;
        const unsigned k = (unsigned)(r % q.nk), n = (unsigned)(r / q.nk);
        double *e = q.p + ((unsigned)q.i0 + i) + ((unsigned)q.j0 + j) * q.sy + k * q.sz + n * q.sn;
        if (b.unpack) *e = q.buf[g]; else q.buf[g] = *e;
    }
}
// local periodic wrap in j of a rank that owns the whole j extent but not the whole i extent: ghost rows over [i0, i0+w)
__global__ void k_wrap_rows(XDesc q, int nrows, int vlo_off, int vhi_off, int leny)
{
    const long w = q.w, total = w * 2 * nrows * q.nk * q.ncomp;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const unsigned i = (unsigned)(g % w); long r = g / w;
        const int row = (int)(r % (2 * nrows)); r /= (2 * nrows);
        const unsigned k = (unsigned)(r % q.nk), n = (unsigned)(r / q.nk);
        // rows below the valid range copy from leny rows up, rows above it from leny rows down (offsets relative to the array's first row)
        const int jd = row < nrows ? vlo_off - nrows + row : vhi_off + 1 + (row - nrows);
        const int js = row < nrows ? jd + leny : jd - leny;
        double *base = q.p + ((unsigned)q.i0 + i) + k * q.sz + n * q.sn;
        base[(unsigned)jd * q.sy] = base[(unsigned)js * q.sy];
    }
}

struct HaloItem { MFV m; int icomp, ncomp; };
static std::vector<HaloItem> &halo_queue(rh_ctx *c) { return *(std::vector<HaloItem> *)c->haloq; }
void *halo_queue_new() { return new std::vector<HaloItem>(); }
void halo_queue_delete(void *q) { delete (std::vector<HaloItem> *)q; }

static int xbuf_reserve(rh_ctx *c, long n)
{
    if (n <= c->xcap) return 0;                                                 // grow the message slots (first calls only)
    RH_HIP(hipStreamSynchronize(c->stream));
    if (c->d_xbuf) RH_HIP(hipFree(c->d_xbuf));
    c->xcap = (n + 4095) / 4096 * 4096; c->d_xbuf = nullptr;
    RH_HIP(hipMalloc((void **)&c->d_xbuf, sizeof(double) * 4 * c->xcap));
    return 0;
}
static int xbatch_go(rh_ctx *c, XBatch &b, long big)
{
    if (!b.n) return 0;
    long nb = (big + 255) / 256; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
    k_xblocks<<<dim3((unsigned)nb, (unsigned)b.n, 1), 256, 0, c->stream>>>(b);
    b.n = 0;
    RH_LAUNCH_CHECK(); return 0;
}

// one phase (d = 0: X, d = 1: Y) of the neighbour exchange for a list of arrays: one pack launch, one message per neighbour, one unpack launch
static int exchange_phase(rh_ctx *c, const std::vector<HaloItem> &items, int d)
{
    std::vector<rh_halo_plan> pl(items.size());
    long n = 0; int peer_lo = -1, peer_hi = -1;
    for (size_t q = 0; q < items.size(); q++) {
        const HaloItem &it = items[q];
        RH_TRY(plan_dir(c->prm, d, it.m.a.lo, it.m.a.hi, it.m.ng, &pl[q]));
        if (pl[q].nrows <= 0) continue;
        const long w = pl[q].i_hi - pl[q].i_lo + 1;
        n += w * pl[q].nrows * (it.m.a.hi[2] - it.m.a.lo[2] + 1) * it.ncomp;
        peer_lo = pl[q].peer_lo; peer_hi = pl[q].peer_hi;                       // the same for every array of this rank
    }
    if (n <= 0 || (peer_lo < 0 && peer_hi < 0)) return 0;
    if (!c->xfn && !c->comm) return rh_fail("halo_fill_boundary: decomposed run without a communicator (call rh_comm_init or rh_comm_set_exchange)");
    RH_TRY(xbuf_reserve(c, n));
    double *slot[4] = {c->d_xbuf, c->d_xbuf + c->xcap, c->d_xbuf + 2 * c->xcap, c->d_xbuf + 3 * c->xcap};   // send_lo, send_hi, recv_lo, recv_hi
    for (int unpack = 0; unpack < 2; unpack++) {
        XBatch b; b.n = 0; b.unpack = unpack; long big = 0, off = 0;
        for (size_t q = 0; q < items.size(); q++) {
            const HaloItem &it = items[q]; const rh_halo_plan &p = pl[q];
            if (p.nrows <= 0) continue;
            const A4 a = mkA(it.m.a, it.icomp);
            const int nk = a.hi2 - a.lo2 + 1;
            const long wo = p.i_hi - p.i_lo + 1, cnt = wo * p.nrows * nk * it.ncomp;
            for (int side = 0; side < 2; side++) {
                if ((side == 0 ? peer_lo : peer_hi) < 0) continue;
                const int first = unpack ? (side == 0 ? p.recv_lo_j : p.recv_hi_j) : (side == 0 ? p.send_lo_j : p.send_hi_j);
                XDesc x; x.p = a.p; x.buf = slot[2 * unpack + side] + off; x.sy = a.sy; x.sz = a.sz; x.sn = a.sn; x.nk = nk; x.ncomp = it.ncomp;
                if (d == 0) { x.i0 = first - a.lo0; x.w = p.nrows; x.j0 = p.i_lo - a.lo1; x.h = (int)wo; }
                else        { x.j0 = first - a.lo1; x.h = p.nrows; x.i0 = p.i_lo - a.lo0; x.w = (int)wo; }
                if (x.i0 < 0 || x.j0 < 0 || x.i0 + x.w > (int)a.sy || x.j0 + x.h > a.hi1 - a.lo1 + 1) return rh_fail("halo exchange: block outside the array");
                b.q[b.n++] = x; big = std::max(big, cnt);
                if (b.n == X_BATCH) RH_TRY(xbatch_go(c, b, big));
            }
            off += cnt;
        }
        RH_TRY(xbatch_go(c, b, big));
        if (!unpack) RH_TRY(comm_exchange(c, peer_lo >= 0 ? n : 0, peer_lo, peer_hi >= 0 ? n : 0, peer_hi));
    }
    return 0;
}

static bool bc_pending_on(rh_ctx *c, const double *p, long n);
// FillBoundary of a list of arrays (each one distinct): local periodic wraps, then the two exchange phases, batched over the list
static int halo_fill_list(rh_ctx *c, const std::vector<HaloItem> &items)
{
    const int npx = c->prm.nranks[0] < 1 ? 1 : c->prm.nranks[0], npy = c->prm.nranks[1] < 1 ? 1 : c->prm.nranks[1];
    const int lenx = c->dom.hi[0] - c->dom.lo[0] + 1, leny = c->dom.hi[1] - c->dom.lo[1] + 1;
    // a direction wraps locally when one rank owns all of it; the j wrap of an i-decomposed run waits for phase X (it carries ghost columns)
    const int perx = (npx == 1) ? c->prm.periodic[0] : 0, pery = (npy == 1 && npx == 1) ? c->prm.periodic[1] : 0;
    PBatch pb; pb.n = 0; pb.perx = perx; pb.pery = pery; pb.lenx = lenx; pb.leny = leny;
    long pbig = 0;
    auto pgo = [&]() -> int {
        if (!pb.n) return 0;
        long nb = (pbig + 255) / 256; if (nb > 4096) nb = 4096;
        k_fill_periodic<<<dim3((unsigned)nb, (unsigned)pb.n, 1), 256, 0, c->stream>>>(pb);
        pb.n = 0; pbig = 0;
        RH_LAUNCH_CHECK(); return 0;
    };
    for (const HaloItem &it : items) {
        const A4 a = mkA(it.m.a); const B3 v = mkB(it.m.valid);
        const long NX = a.hi0 - a.lo0 + 1, NY = a.hi1 - a.lo1 + 1, vx = v.hi0 - v.lo0 + 1, vy = v.hi1 - v.lo1 + 1;
        if (vx <= 0 || vy <= 0 || !bx_contains(bx_of(it.m.a), it.m.valid)) return rh_fail("halo_fill_boundary: valid box not inside array");
        if (perx || pery) {
            const long total = ((NY - vy) * NX + vy * (NX - vx)) * (a.hi2 - a.lo2 + 1) * it.ncomp;
            if (total > 0) {
                PFill &f = pb.q[pb.n++]; f.a = a; f.v = v; f.icomp = it.icomp; f.ncomp = it.ncomp;
                pbig = std::max(pbig, total);
                if (pb.n == P_BATCH) RH_TRY(pgo());
            }
        }
    }
    RH_TRY(pgo());
    if (npx == 1 && npy == 1) return 0;
    if (npx > 1) RH_TRY(exchange_phase(c, items, 0));
    if (npy > 1) RH_TRY(exchange_phase(c, items, 1));
    else if (npx > 1 && c->prm.periodic[1]) {
        for (const HaloItem &it : items) {
            rh_halo_plan p; RH_TRY(plan_dir(c->prm, 1, it.m.a.lo, it.m.a.hi, it.m.ng, &p));
            if (p.nrows <= 0) continue;
            const A4 a = mkA(it.m.a, it.icomp);
            XDesc x; x.p = a.p; x.buf = nullptr; x.sy = a.sy; x.sz = a.sz; x.sn = a.sn; x.nk = a.hi2 - a.lo2 + 1; x.ncomp = it.ncomp;
            x.i0 = p.i_lo - a.lo0; x.w = p.i_hi - p.i_lo + 1; x.j0 = 0; x.h = 0;
            const long total = (long)x.w * 2 * p.nrows * x.nk * x.ncomp;
            long nb = (total + 255) / 256; if (nb > 1024) nb = 1024;
            k_wrap_rows<<<(int)nb, 256, 0, c->stream>>>(x, p.nrows, it.m.valid.lo[1] - a.lo1, it.m.valid.hi[1] - a.lo1, leny);
            RH_LAUNCH_CHECK();
        }
    }
    return 0;
}

// ---- side stream ------------------------------------------------------------------------------------------------
// The scope swaps the context's stream and message slots, so every launch and every message of the fills inside it goes to the
// side stream; one communicator serves both streams, its operations kept in issue order by an event (comm_exchange).
static void async_swap(rh_ctx *c)
{
    std::swap(c->stream, c->xstream); std::swap(c->d_xbuf, c->d_xbuf2); std::swap(c->xcap, c->xcap2);
}
static int side_stream_ensure(rh_ctx *c);
int halo_async_begin(rh_ctx *c)
{
    if (c->no_async || c->async_on) return 0;
    // Built-in RCCL transport: one communicator would be driven from two streams.  The event in comm_exchange keeps its operations in
    // issue order, but that combination has only run through the one-rank self-test so far (the decomposition tests drive the side
    // stream through the exchange hook), so it stays opt-in (RH_ASYNC_RCCL=1) until it has run on a multi-GPU node.
    if (c->comm && !c->async_rccl) return 0;
    RH_TRY(bc_flush(c));
    RH_TRY(halo_async_join(c));                          // one scope in flight at a time
    RH_TRY(side_stream_ensure(c));
    RH_HIP(hipEventRecord(c->ev_fork, c->stream));
    RH_HIP(hipStreamWaitEvent(c->xstream, c->ev_fork, 0));
    async_swap(c); c->async_on = true;
    return 0;
}
int halo_async_end(rh_ctx *c)
{
    if (!c->async_on) return 0;
    const int rc = bc_flush(c);
    hipError_t e = hipEventRecord(c->ev_join, c->stream);           // c->stream is the side stream here
    async_swap(c); c->async_on = false; c->async_pending = true;
    if (rc) return rc;
    RH_HIP(e);
    return 0;
}
static int side_stream_ensure(rh_ctx *c)
{
    if (c->xstream) return 0;
    RH_HIP(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    c->own_xstream = c->xstream;
    RH_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)); RH_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    return 0;
}
int halo_side_enter(rh_ctx *c)
{
    if (c->async_on) return rh_fail("halo_side_enter: a side-stream scope is already open");
    RH_TRY(bc_flush(c));
    RH_TRY(side_stream_ensure(c));
    async_swap(c); c->async_on = true;
    return 0;
}
int halo_side_leave(rh_ctx *c)
{
    if (!c->async_on) return 0;
    const int rc = bc_flush(c);
    async_swap(c); c->async_on = false;
    return rc;
}
int halo_async_join(rh_ctx *c)
{
    RH_TRY(step2d_flush_pending(c));
    if (!c->async_pending) return 0;
    c->async_pending = false;
    RH_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return 0;
}

int halo_flush(rh_ctx *c)
{
    std::vector<HaloItem> &Q = halo_queue(c);
    if (Q.empty()) return 0;
    std::vector<HaloItem> items; items.swap(Q);
    return halo_fill_list(c, items);
}

int halo_fill_boundary(rh_ctx *c, const MFV &m, int icomp, int ncomp)
{
    if (icomp < 0 || icomp + ncomp > m.a.ncomp) return rh_fail("halo_fill_boundary: component range");
    if (a4_size(m.a) >= (1L << 32)) return rh_fail("halo_fill_boundary: array too large for 32-bit indexing");
    std::vector<HaloItem> &Q = halo_queue(c);
    // order: a fill never overtakes queued work on the same array (an earlier fill of it, or BC passes that read its ghost cells)
    bool clash = false;
    for (const HaloItem &q : Q) if (q.m.a.p == m.a.p) clash = true;
    if (clash) { RH_TRY(halo_flush(c)); RH_TRY(bc_flush(c)); }
    else if (bc_pending_on(c, m.a.p, a4_size(m.a))) { RH_TRY(halo_flush(c)); RH_TRY(bc_flush(c)); }
    HaloItem it; it.m = m; it.icomp = icomp; it.ncomp = ncomp;
    Q.push_back(it);
    if (c->bc_defer <= 0) return halo_flush(c);
    return 0;
}

// ---- passes of the physical-BC routines over (small) regions ---------------------------------------------------
// A BC routine is a chain of small passes (sides, then edges, then corners) whose order matters only where one pass reads or
// overwrites what another wrote.  Passes are queued with the level of their deepest hazard (RAW / WAW / WAR against queued
// passes on the same component) and flushed as ONE launch per level, 16 passes per launch: the three FillPatch calls that end
// every barotropic call become 3 launches instead of ~28.
enum { BCP_CONST = 0, BCP_COPY_I = 1, BCP_COPY_J = 2, BCP_CORNER = 3, BCP_FLIP_I = 4, BCP_FLIP_J = 5, BCP_ORL_I = 6, BCP_ORL_J = 7 };
struct BcDesc {
    double *d; const double *msk;            // component base of the destination; mask (k = 0 slab) or null
    unsigned sy, sz, msy;                    // strides of destination and mask
    int lo0, lo1, lo2, mlo0, mlo1;           // array origins
    int r0, r1, r2, nx, ny, nk;              // region
    int mode, si, sj; double val;
    // radiation passes only: the previous time level (component base), the tangential mask of the gradients (or null), and the
    // three lines of the stencil along the normal: B = the line the wave leaves from, B2 = the next one inwards, G = the line of
    // `calc` the outgoing value is blended with
    const double *calc, *mt; unsigned csy, csz, mtsy; int clo0, clo1, clo2, mtlo0, mtlo1, B, B2, G;
};
#define BC_BATCH 16
struct BcBatch { int n; BcDesc q[BC_BATCH]; };
// dest(i,j,k) = val*msk | dest(si,j,k)*msk | dest(i,sj,k)*msk | 0.5*(dest(i,sj,k)+dest(si,j,k)) | +-dest(si-i,j,k)*msk | +-dest(i,sj-j,k)*msk
// | Orlanski radiation along i or j;  msk == null: no mask factor
__global__ void k_bc_pass(BcBatch b)
{
    const BcDesc &q = b.q[blockIdx.y];
    const long nx = q.nx, ny = q.ny;
    const long total = nx * ny * q.nk;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const int i = q.r0 + (int)(g % nx), j = q.r1 + (int)((g / nx) % ny), k = q.r2 + (int)(g / (nx * ny));
        const unsigned kz = (unsigned)(k - q.lo2) * q.sz;
        auto at = [&](int ii, int jj) -> double & { return q.d[(unsigned)(ii - q.lo0) + (unsigned)(jj - q.lo1) * q.sy + kz]; };
        double x, den = 1.0;
        if (q.mode == BCP_CONST) x = q.val;
        else if (q.mode == BCP_COPY_I) x = at(q.si, j);
        else if (q.mode == BCP_COPY_J) x = at(i, q.sj);
        else if (q.mode == BCP_CORNER) x = 0.5 * (at(i, q.sj) + at(q.si, j));
        else if (q.mode == BCP_FLIP_I) { x = at(q.si - i, j); if (q.val < 0.0) x = -x; }
        else if (q.mode == BCP_FLIP_J) { x = at(i, q.sj - j); if (q.val < 0.0) x = -x; }
        else {
            // REMORA_BoundaryConditions_cons.cpp:75-87 and its eleven siblings (cons :92-104,:119-131,:136-148; xvel :69-81,:105-117,
            // :148-159,:172-183; yvel :67-78,:91-102,:127-138,:163-175), which differ only in the lines they address
            const unsigned ckz = (unsigned)(k - q.clo2) * q.csz;
            auto ca = [&](int ii, int jj) -> double { return q.calc[(unsigned)(ii - q.clo0) + (unsigned)(jj - q.clo1) * q.csy + ckz]; };
            const bool ni = (q.mode == BCP_ORL_I);
            const double cB = ni ? ca(q.B, j) : ca(i, q.B);
            double gA = ni ? cB - ca(q.B, j - 1) : cB - ca(i - 1, q.B);
            double gB = ni ? ca(q.B, j + 1) - cB : ca(i + 1, q.B) - cB;
            if (q.mt) { const double m = q.mt[(unsigned)(i - q.mtlo0) + (unsigned)(j - q.mtlo1) * q.mtsy]; gA = gA * m; gB = gB * m; }
            const double dB = ni ? at(q.B, j) : at(i, q.B), dB2 = ni ? at(q.B2, j) : at(i, q.B2);
            double dTdt = cB - dB;
            const double dTdn = dB - dB2;
            if (dTdt * dTdn < 0.0) dTdt = 0.0;
            const double dTan = (dTdt * (gA + gB) > 0.0) ? gA : gB;
            const double s2 = dTdn * dTdn + dTan * dTan;
            const double cff = (s2 < 1.0e-20) ? 1.0e-20 : s2;
            const double Cn = dTdt * dTdn;
            x = cff * (ni ? ca(q.G, j) : ca(i, q.G)) + Cn * dB;
            den = cff + Cn;
        }
        if (q.msk) x = x * q.msk[(unsigned)(i - q.mlo0) + (unsigned)(j - q.mlo1) * q.msy];
        if (q.mode >= BCP_ORL_I) x = x / den;
        at(i, j) = x;
    }
}

struct BcPending { BcDesc q; rh_box w, ra, rb; int level; long npts; };   // write box, up to two read boxes
static inline bool bx_hit(const rh_box &a, const rh_box &b) { return bx_ok(a) && bx_ok(b) && bx_ok(bx_and(a, b)); }

void *bc_queue_new() { return new std::vector<BcPending>(); }
void bc_queue_delete(void *q) { delete (std::vector<BcPending> *)q; }

int bc_flush(rh_ctx *c)
{
    RH_TRY(halo_flush(c));                                // queued ghost exchanges land before the BC passes that read them
    std::vector<BcPending> &Q = *(std::vector<BcPending> *)c->bcq;
    if (Q.empty()) return 0;
    int maxl = 0; for (const BcPending &p : Q) maxl = std::max(maxl, p.level);
    for (int l = 0; l <= maxl; l++) {
        BcBatch b; b.n = 0; long big = 0;
        auto go = [&]() -> int {
            if (!b.n) return 0;
            long nb = (big + 255) / 256; if (nb > 256) nb = 256; if (nb < 1) nb = 1;
            k_bc_pass<<<dim3((unsigned)nb, (unsigned)b.n, 1), 256, 0, c->stream>>>(b);
            b.n = 0; big = 0;
            RH_LAUNCH_CHECK(); return 0;
        };
        for (const BcPending &p : Q) if (p.level == l) { b.q[b.n++] = p.q; big = std::max(big, p.npts); if (b.n == BC_BATCH) RH_TRY(go()); }
        RH_TRY(go());
    }
    Q.clear();
    return 0;
}
// defer passes until the matching bc_end (nesting allowed); only pure ghost-fill sequences may sit inside such a scope
void bc_begin(rh_ctx *c) { c->bc_defer++; }
int  bc_end(rh_ctx *c) { if (--c->bc_defer <= 0) { c->bc_defer = 0; return bc_flush(c); } return 0; }
// queued passes on [p, p + n) must land before anything else touches that memory
static bool bc_pending_on(rh_ctx *c, const double *p, long n)
{
    std::vector<BcPending> &Q = *(std::vector<BcPending> *)c->bcq;
    for (const BcPending &q : Q) if (q.q.d >= p && q.q.d < p + n) return true;
    return false;
}

// what a radiation pass reads besides the destination
struct OrlArgs { const rh_array4 *calc; int ccomp; const rh_array4 *mt; int B, B2, G; };

static int bc_pass(rh_ctx *c, const rh_array4 &dest, int comp, rh_box r, int mode, int si, int sj, double val, const rh_array4 *msk,
                   const OrlArgs *orl = nullptr)
{
    r = bx_and(r, bx_of(dest));
    if (!bx_ok(r)) return 0;
    A4 d = mkA(dest, comp);
    if (a4_size(dest) >= (1L << 32)) return rh_fail("bc_pass: array too large for 32-bit indexing");
    // every source index must lie inside the array
    if ((mode == BCP_COPY_I || mode == BCP_CORNER) && (si < dest.lo[0] || si > dest.hi[0])) return rh_fail("bc_pass: source i outside array");
    if ((mode == BCP_COPY_J || mode == BCP_CORNER) && (sj < dest.lo[1] || sj > dest.hi[1])) return rh_fail("bc_pass: source j outside array");
    if (mode == BCP_FLIP_I && (si - r.hi[0] < dest.lo[0] || si - r.lo[0] > dest.hi[0])) return rh_fail("bc_pass: mirrored i outside array");
    if (mode == BCP_FLIP_J && (sj - r.hi[1] < dest.lo[1] || sj - r.lo[1] > dest.hi[1])) return rh_fail("bc_pass: mirrored j outside array");
    BcPending p; memset(&p, 0, sizeof p);
    p.q.d = d.p; p.q.sy = d.sy; p.q.sz = d.sz; p.q.lo0 = d.lo0; p.q.lo1 = d.lo1; p.q.lo2 = d.lo2;
    if (msk) {
        rh_box r2 = bx_slab(r, 2, 0);
        if (!bx_contains(bx_of(*msk), r2)) return rh_fail("bc_pass: mask does not cover the region");
        p.q.msk = msk->p; p.q.msy = (unsigned)(msk->hi[0] - msk->lo[0] + 1); p.q.mlo0 = msk->lo[0]; p.q.mlo1 = msk->lo[1];
    }
    p.q.r0 = r.lo[0]; p.q.r1 = r.lo[1]; p.q.r2 = r.lo[2]; p.q.nx = r.hi[0] - r.lo[0] + 1; p.q.ny = r.hi[1] - r.lo[1] + 1; p.q.nk = r.hi[2] - r.lo[2] + 1;
    p.q.mode = mode; p.q.si = si; p.q.sj = sj; p.q.val = val;
    p.w = r; p.npts = bx_npts(r);
    const rh_box none = bx_make(0, 0, 0, -1, -1, -1);
    p.ra = none; p.rb = none;
    if (mode == BCP_COPY_I || mode == BCP_CORNER) { p.ra = r; p.ra.lo[0] = p.ra.hi[0] = si; }
    if (mode == BCP_COPY_J || mode == BCP_CORNER) { p.rb = r; p.rb.lo[1] = p.rb.hi[1] = sj; }
    if (mode == BCP_FLIP_I) { p.ra = r; p.ra.lo[0] = si - r.hi[0]; p.ra.hi[0] = si - r.lo[0]; }
    if (mode == BCP_FLIP_J) { p.ra = r; p.ra.lo[1] = sj - r.hi[1]; p.ra.hi[1] = sj - r.lo[1]; }
    if (mode == BCP_ORL_I || mode == BCP_ORL_J) {
        if (!orl || !orl->calc || !orl->calc->p)
            return rh_fail("physbc: an orlanski_rad side needs the calc array (the reference reads an empty Array4 here; use rh_physbc_calc / rh_fill_patch_calc)");
        if (!msk) return rh_fail("bc_pass: radiation pass without a mask");
        const int nd = (mode == BCP_ORL_I) ? 0 : 1, td = 1 - nd;
        const rh_array4 &ca = *orl->calc;
        if (orl->ccomp < 0 || orl->ccomp >= ca.ncomp) return rh_fail("bc_pass: calc component");
        if (a4_size(ca) >= (1L << 32)) return rh_fail("bc_pass: calc array too large for 32-bit indexing");
        // the stencil: lines B, B2 of the destination; lines B (one more point either way along the side) and G of calc
        rh_box sd = r; sd.lo[nd] = std::min(orl->B, orl->B2); sd.hi[nd] = std::max(orl->B, orl->B2);
        if (!bx_contains(bx_of(dest), sd)) return rh_fail("bc_pass: radiation stencil outside the array");
        rh_box sc = r; sc.lo[nd] = std::min(orl->B, orl->G); sc.hi[nd] = std::max(orl->B, orl->G); sc.lo[td] -= 1; sc.hi[td] += 1;
        if (!bx_contains(bx_of(ca), sc)) return rh_fail("bc_pass: radiation stencil outside the calc array");
        A4 cv = mkA(ca, orl->ccomp);
        p.q.calc = cv.p; p.q.csy = cv.sy; p.q.csz = cv.sz; p.q.clo0 = cv.lo0; p.q.clo1 = cv.lo1; p.q.clo2 = cv.lo2;
        if (orl->mt) {
            if (!bx_contains(bx_of(*orl->mt), bx_slab(r, 2, 0))) return rh_fail("bc_pass: tangential mask does not cover the region");
            p.q.mt = orl->mt->p; p.q.mtsy = (unsigned)(orl->mt->hi[0] - orl->mt->lo[0] + 1); p.q.mtlo0 = orl->mt->lo[0]; p.q.mtlo1 = orl->mt->lo[1];
        }
        p.q.B = orl->B; p.q.B2 = orl->B2; p.q.G = orl->G;
        p.ra = sd;
        if (cv.p == d.p) {
            // calc is the destination itself (REMORA::InitData): its reads are hazards like any other.  A pass must not read what it
            // writes except cell by cell: the caller splits the side so that the G line is a pass of its own
            p.rb = sc;
            rh_box gl = r; gl.lo[nd] = gl.hi[nd] = orl->G;
            if (bx_ok(bx_and(r, gl)) && (r.lo[nd] != r.hi[nd])) return rh_fail("bc_pass: in-place radiation pass covers its own G line");
        } else if (bc_pending_on(c, ca.p, a4_size(ca))) RH_TRY(bc_flush(c));
    }
    // the other direction: a queued radiation pass READS a calc array that is not its destination; a pass that writes into that array
    // must not share its launch level (the level loop below only orders passes on one destination).  No call site of the drivers does
    // this (calc is always an *_old array nobody fills in the same scope); the queue is flushed first if one ever does.
    {
        std::vector<BcPending> &Q0 = *(std::vector<BcPending> *)c->bcq;
        const double *lo = dest.p, *hi = dest.p + a4_size(dest);
        bool clash = false;
        for (const BcPending &o : Q0) if (o.q.calc && o.q.calc != o.q.d && o.q.calc >= lo && o.q.calc < hi) { clash = true; break; }
        if (clash) RH_TRY(bc_flush(c));
    }
    std::vector<BcPending> &Q = *(std::vector<BcPending> *)c->bcq;
    p.level = 0;
    for (const BcPending &o : Q) {
        if (o.q.d != p.q.d) continue;
        if (bx_hit(o.w, p.ra) || bx_hit(o.w, p.rb) || bx_hit(o.w, p.w) || bx_hit(o.ra, p.w) || bx_hit(o.rb, p.w)) p.level = std::max(p.level, o.level + 1);
    }
    Q.push_back(p);
    if (c->bc_defer <= 0) return bc_flush(c);
    return 0;
}
// one side of a radiation condition over region r (all its ghost lines).  With calc == dest the lines are issued in the order the
// reference's CPU loops visit them (ascending index): below the domain the far lines read the old G line before it is
// overwritten, above the domain the G line comes first and the far lines read its new value.
static int bc_orlanski(rh_ctx *c, const rh_array4 &dest, int comp, rh_box r, int nd, int hi, const rh_array4 *msk, const OrlArgs &o)
{
    const int mode = nd == 0 ? BCP_ORL_I : BCP_ORL_J;
    const bool inplace = o.calc && o.calc->p && mkA(*o.calc, o.ccomp).p == mkA(dest, comp).p;
    rh_box rr = bx_and(r, bx_of(dest));
    if (!inplace || !bx_ok(rr) || rr.lo[nd] == rr.hi[nd]) return bc_pass(c, dest, comp, r, mode, 0, 0, 0.0, msk, &o);
    rh_box gl = rr; gl.lo[nd] = gl.hi[nd] = o.G;
    rh_box far = rr; if (hi) far.lo[nd] = o.G + 1; else far.hi[nd] = o.G - 1;
    if (!bx_ok(bx_and(rr, gl))) return bc_pass(c, dest, comp, r, mode, 0, 0, 0.0, msk, &o);      // G is not a ghost line of this side (velocity faces)
    if (hi) { RH_TRY(bc_pass(c, dest, comp, gl, mode, 0, 0, 0.0, msk, &o)); return bc_pass(c, dest, comp, far, mode, 0, 0, 0.0, msk, &o); }
    RH_TRY(bc_pass(c, dest, comp, far, mode, 0, 0, 0.0, msk, &o)); return bc_pass(c, dest, comp, gl, mode, 0, 0, 0.0, msk, &o);
}

// amrex::setBC: domain type where the box reaches the domain side, int_dir otherwise
static BCRec3 setbc(const BCRec3 &d, const rh_box &bx, const rh_box &dom)
{
    BCRec3 r;
    for (int k = 0; k < 3; k++) { r.lo[k] = bx.lo[k] <= dom.lo[k] ? d.lo[k] : RH_BC_INT_DIR; r.hi[k] = bx.hi[k] >= dom.hi[k] ? d.hi[k] : RH_BC_INT_DIR; }
    return r;
}
// nodal-in-z arrays: init_bcs gives zvel_bc foextrap on every face (:311-316); anything else cannot reach impose_zvel
static int supported(const BCRec3 &bc)
{
    for (int d = 0; d < 3; d++) for (int t : {bc.lo[d], bc.hi[d]})
        if (t != RH_BC_INT_DIR && t != RH_BC_FOEXTRAP && t != RH_BC_EXT_DIR && t != -666) return rh_fail("physbc: BC type %d on a w-point array", t);
    return 0;
}

// what REMORAPhysBCFunct::operator() takes besides the array and its mask (PhysBCFunct.cpp:21-23); null = the empty MultiFab
struct BcAux { const rh_array4 *calc, *msku, *mskv; };
static inline bool copy_like_cons(int t) { return t == RH_BC_FOEXTRAP || t == RH_BC_CLAMPED || t == RH_BC_CHAPMAN || t == RH_BC_ORLANSKI_RAD || t == RH_BC_ORLANSKI_RAD_NUDGE; }
static inline bool skip_corner(int t) { return t == RH_BC_CLAMPED || t == RH_BC_FLATHER || t == RH_BC_CHAPMAN || t == RH_BC_ORLANSKI_RAD_NUDGE; }

// REMORA_BoundaryConditions_cons.cpp:20-340
static int impose_cons(rh_ctx *c, const MFV &m, const rh_array4 &mskr, int icomp, int ncomp, int bccomp, int nnf, const BcAux &aux)
{
    const rh_box dom = c->dom, bx = bx_of(m.a), vb = m.valid;
    const int perx = c->prm.periodic[0], pery = c->prm.periodic[1];
    const int dlx = dom.lo[0], dhx = dom.hi[0], dly = dom.lo[1], dhy = dom.hi[1];
    for (int n = 0; n < ncomp; n++) {
        const BCRec3 bc = setbc(c->bcs[bccomp + n], bx, dom);                                    // :33-34
        const double *ext = c->extdir[bccomp + n];
        const int cc = icomp + n;
        if (!perx) {                                                                             // :65-107
            rh_box lo = bx; lo.hi[0] = dlx - 1;
            lo.lo[1] = std::max(vb.lo[1] - 1, dly); lo.hi[1] = std::min(vb.hi[1] + 1, dhy);
            rh_box hi = bx; hi.lo[0] = dhx + 1; hi.lo[1] = lo.lo[1]; hi.hi[1] = lo.hi[1];
            if (bc.lo[0] == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, cc, lo, BCP_CONST, 0, 0, ext[0], &mskr));
            else if (bc.lo[0] == RH_BC_ORLANSKI_RAD) { const OrlArgs o = {aux.calc, cc, aux.mskv, dlx, dlx + 1, dlx - 1}; RH_TRY(bc_orlanski(c, m.a, cc, lo, 0, 0, &mskr, o)); }
            if (bc.hi[0] == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, cc, hi, BCP_CONST, 0, 0, ext[3], &mskr));
            else if (bc.hi[0] == RH_BC_ORLANSKI_RAD) { const OrlArgs o = {aux.calc, cc, aux.mskv, dhx, dhx - 1, dhx + 1}; RH_TRY(bc_orlanski(c, m.a, cc, hi, 0, 1, &mskr, o)); }
        }
        if (!pery) {                                                                             // :109-151
            rh_box lo = bx; lo.hi[1] = dly - 1;
            lo.lo[0] = std::max(vb.lo[0] - 1, dlx); lo.hi[0] = std::min(vb.hi[0] + 1, dhx);
            rh_box hi = bx; hi.lo[1] = dhy + 1; hi.lo[0] = lo.lo[0]; hi.hi[0] = lo.hi[0];
            if (bc.lo[1] == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, cc, lo, BCP_CONST, 0, 0, ext[1], &mskr));
            else if (bc.lo[1] == RH_BC_ORLANSKI_RAD) { const OrlArgs o = {aux.calc, cc, aux.msku, dly, dly + 1, dly - 1}; RH_TRY(bc_orlanski(c, m.a, cc, lo, 1, 0, &mskr, o)); }
            if (bc.hi[1] == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, cc, hi, BCP_CONST, 0, 0, ext[4], &mskr));
            else if (bc.hi[1] == RH_BC_ORLANSKI_RAD) { const OrlArgs o = {aux.calc, cc, aux.msku, dhy, dhy - 1, dhy + 1}; RH_TRY(bc_orlanski(c, m.a, cc, hi, 1, 1, &mskr, o)); }
        }
        // z sides (:153-168, :254-286): the arrays on this path carry no ghost cells in z, nothing to fill
        const int zl = std::max(dom.lo[2], bx.lo[2]), zh = std::min(dom.hi[2], bx.hi[2]);
        // 2-D arrays sit on the k = 0 slab whatever the domain's k range
        const int zlo = (bx.lo[2] == bx.hi[2]) ? bx.lo[2] : zl, zhi = (bx.lo[2] == bx.hi[2]) ? bx.hi[2] : zh;
        rh_box xlo = bx; xlo.hi[0] = dlx - 1 - nnf; xlo.lo[2] = zlo; xlo.hi[2] = zhi;            // :170-181
        rh_box xhi = bx; xhi.lo[0] = dhx + 1 + nnf; xhi.lo[2] = zlo; xhi.hi[2] = zhi;
        rh_box ylo = bx; ylo.hi[1] = dly - 1 - nnf; ylo.lo[2] = zlo; ylo.hi[2] = zhi;
        rh_box yhi = bx; yhi.lo[1] = dhy + 1 + nnf; yhi.lo[2] = zlo; yhi.hi[2] = zhi;
        if (!perx || bccomp == RH_BCV_FOEXTRAP) {                                                // :193-221
            const int il = (bc.lo[0] == RH_BC_ORLANSKI_RAD) ? 1 : 0, ih = (bc.hi[0] == RH_BC_ORLANSKI_RAD) ? 1 : 0;
            if (copy_like_cons(bc.lo[0])) RH_TRY(bc_pass(c, m.a, cc, xlo, BCP_COPY_I, dlx - nnf - il, 0, 0.0, nullptr));
            else if (bc.lo[0] == RH_BC_REFLECT_EVEN || bc.lo[0] == RH_BC_REFLECT_ODD) RH_TRY(bc_pass(c, m.a, cc, xlo, BCP_FLIP_I, dlx - 1, 0, bc.lo[0] == RH_BC_REFLECT_ODD ? -1.0 : 1.0, nullptr));
            if (copy_like_cons(bc.hi[0])) RH_TRY(bc_pass(c, m.a, cc, xhi, BCP_COPY_I, dhx + nnf + ih, 0, 0.0, nullptr));
            else if (bc.hi[0] == RH_BC_REFLECT_EVEN || bc.hi[0] == RH_BC_REFLECT_ODD) RH_TRY(bc_pass(c, m.a, cc, xhi, BCP_FLIP_I, 2 * dhx + 1, 0, bc.hi[0] == RH_BC_REFLECT_ODD ? -1.0 : 1.0, nullptr));
        }
        if (!pery || bccomp == RH_BCV_FOEXTRAP) {                                                // :223-252
            const int il = (bc.lo[1] == RH_BC_ORLANSKI_RAD) ? 1 : 0, ih = (bc.hi[1] == RH_BC_ORLANSKI_RAD) ? 1 : 0;
            if (copy_like_cons(bc.lo[1])) RH_TRY(bc_pass(c, m.a, cc, ylo, BCP_COPY_J, 0, dly - nnf - il, 0.0, nullptr));
            else if (bc.lo[1] == RH_BC_REFLECT_EVEN || bc.lo[1] == RH_BC_REFLECT_ODD) RH_TRY(bc_pass(c, m.a, cc, ylo, BCP_FLIP_J, 0, dly - 1, bc.lo[1] == RH_BC_REFLECT_ODD ? -1.0 : 1.0, nullptr));
            // :242 asks the LOW side's type for orlanski_rad (sic)
            if (bc.hi[1] == RH_BC_FOEXTRAP || bc.hi[1] == RH_BC_CLAMPED || bc.hi[1] == RH_BC_CHAPMAN || bc.lo[1] == RH_BC_ORLANSKI_RAD || bc.hi[1] == RH_BC_ORLANSKI_RAD_NUDGE)
                RH_TRY(bc_pass(c, m.a, cc, yhi, BCP_COPY_J, 0, dhy + nnf + ih, 0.0, nullptr));
            else if (bc.hi[1] == RH_BC_REFLECT_EVEN || bc.hi[1] == RH_BC_REFLECT_ODD) RH_TRY(bc_pass(c, m.a, cc, yhi, BCP_FLIP_J, 0, 2 * dhy + 1, bc.hi[1] == RH_BC_REFLECT_ODD ? -1.0 : 1.0, nullptr));
        }
        if ((!perx && !pery) || bccomp == RH_BCV_FOEXTRAP) {                                     // :287-337
            if (!skip_corner(bc.lo[0]) && !skip_corner(bc.lo[1])) RH_TRY(bc_pass(c, m.a, cc, bx_and(xlo, ylo), BCP_CORNER, dlx, dly, 0.0, nullptr));
            if (!skip_corner(bc.lo[0]) && !skip_corner(bc.hi[1])) RH_TRY(bc_pass(c, m.a, cc, bx_and(xlo, yhi), BCP_CORNER, dlx, dhy, 0.0, nullptr));
            if (!skip_corner(bc.hi[0]) && !skip_corner(bc.lo[1])) RH_TRY(bc_pass(c, m.a, cc, bx_and(xhi, ylo), BCP_CORNER, dhx, dly, 0.0, nullptr));
            if (!skip_corner(bc.hi[0]) && !skip_corner(bc.hi[1])) RH_TRY(bc_pass(c, m.a, cc, bx_and(xhi, yhi), BCP_CORNER, dhx, dhy, 0.0, nullptr));
        }
    }
    return 0;
}

// one side of a face-centred array, tangential to the faces (xvel :136-191, yvel :55-110): dir = the side's direction
static int vel_tangential_side(rh_ctx *c, const MFV &m, const rh_array4 &msk, int comp, int dir, int hi, int type, double ext,
                               rh_box r, int dl, int dh, const BcAux &aux)
{
    const int B = hi ? dh : dl, B2 = hi ? dh - 1 : dl + 1, G = hi ? dh + 1 : dl - 1;
    const int cmode = dir == 0 ? BCP_COPY_I : BCP_COPY_J, fmode = dir == 0 ? BCP_FLIP_I : BCP_FLIP_J;
    const int flip = hi ? 2 * dh + 1 : dl - 1;
    if (type == RH_BC_EXT_DIR) return bc_pass(c, m.a, comp, r, BCP_CONST, 0, 0, ext, &msk);
    if (type == RH_BC_FOEXTRAP || type == RH_BC_CLAMPED) return bc_pass(c, m.a, comp, r, cmode, B, B, 0.0, &msk);
    if (type == RH_BC_ORLANSKI_RAD) { const OrlArgs o = {aux.calc, 0, nullptr, B, B2, G}; return bc_orlanski(c, m.a, comp, r, dir, hi, &msk, o); }
    if (type == RH_BC_REFLECT_EVEN || type == RH_BC_REFLECT_ODD) return bc_pass(c, m.a, comp, r, fmode, flip, flip, type == RH_BC_REFLECT_ODD ? -1.0 : 1.0, &msk);
    return 0;
}
// one side of a face-centred array, normal to the faces (xvel :56-134, yvel :112-192): the face line, then the ghost lines
static int vel_normal_side(rh_ctx *c, const MFV &m, const rh_array4 &msk, int comp, int dir, int hi, int type, double ext,
                           rh_box face, rh_box ghost, int dl, int dh, const BcAux &aux)
{
    const int cmode = dir == 0 ? BCP_COPY_I : BCP_COPY_J, fmode = dir == 0 ? BCP_FLIP_I : BCP_FLIP_J;
    const int F = hi ? dh + 1 : dl;                       // the boundary face
    const int in1 = hi ? dh : dl + 1, in2 = hi ? dh - 1 : dl + 2;
    // the face itself: only ext_dir, actual outflow and radiation set it
    if (type == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, comp, face, BCP_CONST, 0, 0, ext, &msk));
    else if (type == RH_BC_FOEXTRAP) RH_TRY(bc_pass(c, m.a, comp, face, cmode, in1, in1, 0.0, &msk));
    else if (type == RH_BC_ORLANSKI_RAD) {
        // low side (:69-81): the wave leaves from the first interior face, blended with calc ON the boundary face;
        // high side (:105-117): written like the cell-centred form, from face dom_hi (interior) towards dom_hi + 1
        const OrlArgs o = {aux.calc, 0, nullptr, in1, in2, F};
        RH_TRY(bc_orlanski(c, m.a, comp, face, dir, hi, &msk, o));
    }
    if (type == RH_BC_EXT_DIR) return bc_pass(c, m.a, comp, ghost, BCP_CONST, 0, 0, ext, &msk);
    if (type == RH_BC_FOEXTRAP) return bc_pass(c, m.a, comp, ghost, cmode, in1, in1, 0.0, &msk);                       // inner = 1
    if (type == RH_BC_CLAMPED || type == RH_BC_ORLANSKI_RAD) return bc_pass(c, m.a, comp, ghost, cmode, F, F, 0.0, &msk);   // inner = 0
    if (type == RH_BC_REFLECT_EVEN || type == RH_BC_REFLECT_ODD) { const int fl = hi ? 2 * F : F; return bc_pass(c, m.a, comp, ghost, fmode, fl, fl, type == RH_BC_REFLECT_ODD ? -1.0 : 1.0, &msk); }   // :86,:121: dom_lo - i below, 2 (dom_hi + 1) - i above
    return 0;
}

// REMORA_BoundaryConditions_xvel.cpp:12-269
static int impose_xvel(rh_ctx *c, const MFV &m, const rh_array4 &msku, int comp, int bccomp, const BcAux &aux)
{
    const rh_box dom = c->dom, bx = bx_of(m.a);
    const int perx = c->prm.periodic[0], pery = c->prm.periodic[1];
    const BCRec3 bc = setbc(c->bcs[bccomp], bx, dom); const double *ext = c->extdir[bccomp];
    const bool fx = (bccomp == RH_BCV_FOEXTRAP);
    if (!perx || fx) {                                                                           // :56-134
        rh_box xlo = bx; xlo.hi[0] = dom.lo[0] - 1; rh_box xhi = bx; xhi.lo[0] = dom.hi[0] + 2;
        rh_box flo = bx; flo.lo[0] = flo.hi[0] = dom.lo[0]; rh_box fhi = bx; fhi.lo[0] = fhi.hi[0] = dom.hi[0] + 1;
        xlo = bx_grow(xlo, 0, -1, 0); xhi = bx_grow(xhi, 0, -1, 0); flo = bx_grow(flo, 0, -1, 0); fhi = bx_grow(fhi, 0, -1, 0);
        RH_TRY(vel_normal_side(c, m, msku, comp, 0, 0, bc.lo[0], ext[0], flo, xlo, dom.lo[0], dom.hi[0], aux));
        RH_TRY(vel_normal_side(c, m, msku, comp, 0, 1, bc.hi[0], ext[3], fhi, xhi, dom.lo[0], dom.hi[0], aux));
    }
    if (!pery || fx) {                                                                           // :136-191
        rh_box ylo = bx; ylo.hi[1] = dom.lo[1] - 1; rh_box yhi = bx; yhi.lo[1] = dom.hi[1] + 1;
        ylo = bx_grow(ylo, -1, 0, 0); yhi = bx_grow(yhi, -1, 0, 0);
        RH_TRY(vel_tangential_side(c, m, msku, comp, 1, 0, bc.lo[1], ext[1], ylo, dom.lo[1], dom.hi[1], aux));
        RH_TRY(vel_tangential_side(c, m, msku, comp, 1, 1, bc.hi[1], ext[4], yhi, dom.lo[1], dom.hi[1], aux));
    }
    // z sides (:193-223): no ghost cells in z on this path
    if ((!perx || fx) && (!pery || fx)) {                                                        // :225-265
        rh_box xlo = bx; xlo.hi[0] = dom.lo[0]; rh_box xhi = bx; xhi.lo[0] = dom.hi[0] + 1;
        rh_box ylo = bx; ylo.hi[1] = dom.lo[1] - 1; rh_box yhi = bx; yhi.lo[1] = dom.hi[1] + 1;
        // :236-239 ask the DOMAIN's types, not the box's
        const BCRec3 &db = c->bcs[bccomp];
        const bool cw = db.lo[0] == RH_BC_CLAMPED, ce = db.hi[0] == RH_BC_CLAMPED, cs = db.lo[1] == RH_BC_CLAMPED, cn = db.hi[1] == RH_BC_CLAMPED;
        if (!cw && !cs) RH_TRY(bc_pass(c, m.a, comp, bx_and(xlo, ylo), BCP_CORNER, dom.lo[0] + 1, dom.lo[1], 0.0, nullptr));
        if (!cw && !cn) RH_TRY(bc_pass(c, m.a, comp, bx_and(xlo, yhi), BCP_CORNER, dom.lo[0] + 1, dom.hi[1], 0.0, nullptr));
        if (!ce && !cs) RH_TRY(bc_pass(c, m.a, comp, bx_and(xhi, ylo), BCP_CORNER, dom.hi[0], dom.lo[1], 0.0, nullptr));
        if (!ce && !cn) RH_TRY(bc_pass(c, m.a, comp, bx_and(xhi, yhi), BCP_CORNER, dom.hi[0], dom.hi[1], 0.0, nullptr));
    }
    return 0;
}

// REMORA_BoundaryConditions_yvel.cpp:12-268
static int impose_yvel(rh_ctx *c, const MFV &m, const rh_array4 &mskv, int comp, int bccomp, const BcAux &aux)
{
    const rh_box dom = c->dom, bx = bx_of(m.a);
    const int perx = c->prm.periodic[0], pery = c->prm.periodic[1];
    const BCRec3 bc = setbc(c->bcs[bccomp], bx, dom); const double *ext = c->extdir[bccomp];
    const bool fx = (bccomp == RH_BCV_FOEXTRAP);
    if (!perx || fx) {                                                                           // :55-110
        rh_box xlo = bx; xlo.hi[0] = dom.lo[0] - 1; rh_box xhi = bx; xhi.lo[0] = dom.hi[0] + 1;
        xlo = bx_grow(xlo, 0, -1, 0); xhi = bx_grow(xhi, 0, -1, 0);
        RH_TRY(vel_tangential_side(c, m, mskv, comp, 0, 0, bc.lo[0], ext[0], xlo, dom.lo[0], dom.hi[0], aux));
        RH_TRY(vel_tangential_side(c, m, mskv, comp, 0, 1, bc.hi[0], ext[3], xhi, dom.lo[0], dom.hi[0], aux));
    }
    if (!pery || fx) {                                                                           // :112-192
        rh_box ylo = bx; ylo.hi[1] = dom.lo[1] - 1; rh_box yhi = bx; yhi.lo[1] = dom.hi[1] + 2;
        rh_box flo = bx; flo.lo[1] = flo.hi[1] = dom.lo[1]; rh_box fhi = bx; fhi.lo[1] = fhi.hi[1] = dom.hi[1] + 1;
        ylo = bx_grow(ylo, -1, 0, 0); yhi = bx_grow(yhi, -1, 0, 0); flo = bx_grow(flo, -1, 0, 0); fhi = bx_grow(fhi, -1, 0, 0);
        RH_TRY(vel_normal_side(c, m, mskv, comp, 1, 0, bc.lo[1], ext[1], flo, ylo, dom.lo[1], dom.hi[1], aux));
        RH_TRY(vel_normal_side(c, m, mskv, comp, 1, 1, bc.hi[1], ext[4], fhi, yhi, dom.lo[1], dom.hi[1], aux));
    }
    if ((!perx || fx) && (!pery || fx)) {                                                        // :226-265
        rh_box xlo = bx; xlo.hi[0] = dom.lo[0] - 1; rh_box xhi = bx; xhi.lo[0] = dom.hi[0] + 1;
        rh_box ylo = bx; ylo.hi[1] = dom.lo[1]; rh_box yhi = bx; yhi.lo[1] = dom.hi[1] + 1;
        const BCRec3 &db = c->bcs[bccomp];
        const bool cw = db.lo[0] == RH_BC_CLAMPED, ce = db.hi[0] == RH_BC_CLAMPED, cs = db.lo[1] == RH_BC_CLAMPED, cn = db.hi[1] == RH_BC_CLAMPED;
        if (!cw && !cs) RH_TRY(bc_pass(c, m.a, comp, bx_and(xlo, ylo), BCP_CORNER, dom.lo[0], dom.lo[1] + 1, 0.0, nullptr));
        if (!cw && !cn) RH_TRY(bc_pass(c, m.a, comp, bx_and(xlo, yhi), BCP_CORNER, dom.lo[0], dom.hi[1], 0.0, nullptr));
        if (!ce && !cs) RH_TRY(bc_pass(c, m.a, comp, bx_and(xhi, ylo), BCP_CORNER, dom.hi[0], dom.lo[1] + 1, 0.0, nullptr));
        if (!ce && !cn) RH_TRY(bc_pass(c, m.a, comp, bx_and(xhi, yhi), BCP_CORNER, dom.hi[0], dom.hi[1], 0.0, nullptr));
    }
    return 0;
}

// REMORA_BoundaryConditions_zvel.cpp:13-124
static int impose_zvel(rh_ctx *c, const MFV &m, int comp, int bccomp)
{
    const rh_box dom = c->dom, bx = bx_of(m.a);
    const int perx = c->prm.periodic[0], pery = c->prm.periodic[1];
    const BCRec3 bc = setbc(c->bcs[bccomp], bx, dom); const double *ext = c->extdir[bccomp];
    RH_TRY(supported(bc));
    const bool fx = (bccomp == RH_BCV_FOEXTRAP);
    if (!perx || fx) {
        rh_box xlo = bx; xlo.hi[0] = dom.lo[0] - 1; rh_box xhi = bx; xhi.lo[0] = dom.hi[0] + 1;
        if (bc.lo[0] == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, comp, xlo, BCP_CONST, 0, 0, ext[0], nullptr));
        else if (bc.lo[0] == RH_BC_FOEXTRAP) RH_TRY(bc_pass(c, m.a, comp, xlo, BCP_COPY_I, dom.lo[0], 0, 0.0, nullptr));
        if (bc.hi[0] == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, comp, xhi, BCP_CONST, 0, 0, ext[3], nullptr));
        else if (bc.hi[0] == RH_BC_FOEXTRAP) RH_TRY(bc_pass(c, m.a, comp, xhi, BCP_COPY_I, dom.hi[0], 0, 0.0, nullptr));
    }
    if (!pery || fx) {
        rh_box ylo = bx; ylo.hi[1] = dom.lo[1] - 1; rh_box yhi = bx; yhi.lo[1] = dom.hi[1] + 1;
        if (bc.lo[1] == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, comp, ylo, BCP_CONST, 0, 0, ext[1], nullptr));
        else if (bc.lo[1] == RH_BC_FOEXTRAP) RH_TRY(bc_pass(c, m.a, comp, ylo, BCP_COPY_J, 0, dom.lo[1], 0.0, nullptr));
        if (bc.hi[1] == RH_BC_EXT_DIR) RH_TRY(bc_pass(c, m.a, comp, yhi, BCP_CONST, 0, 0, ext[4], nullptr));
        else if (bc.hi[1] == RH_BC_FOEXTRAP) RH_TRY(bc_pass(c, m.a, comp, yhi, BCP_COPY_J, 0, dom.hi[1], 0.0, nullptr));
    }
    return 0;
}

// REMORA_PhysBCFunct.cpp:21-93
static int physbc_impl(rh_ctx *c, const MFV &m, const rh_array4 &msk, int icomp, int ncomp, int bccomp, int n_not_fill, const BcAux &aux)
{
    rh_box gdom = c->dom;
    for (int d = 0; d < 3; d++) if (m.ix[d]) gdom.hi[d] += 1;
    for (int d = 0; d < 2; d++) if (c->prm.periodic[d] && bccomp != RH_BCV_FOEXTRAP) { gdom.lo[d] -= m.ng[d]; gdom.hi[d] += m.ng[d]; }
    rh_box bx = bx_of(m.a);
    // only x-y extents decide (2-D arrays live on k = 0; no z ghost cells on this path)
    gdom.lo[2] = bx.lo[2]; gdom.hi[2] = bx.hi[2];
    if (bx_contains(gdom, bx)) return 0;
    if (icomp < 0 || icomp + ncomp > m.a.ncomp) return rh_fail("halo_physbc: component range");
    if (!m.ix[0] && !m.ix[1] && !m.ix[2]) return impose_cons(c, m, msk, icomp, ncomp, bccomp, n_not_fill, aux);
    if (ncomp != 1) return rh_fail("halo_physbc: velocity BCs take one component");
    if (m.ix[0]) return impose_xvel(c, m, msk, icomp, bccomp, aux);
    if (m.ix[1]) return impose_yvel(c, m, msk, icomp, bccomp, aux);
    return impose_zvel(c, m, icomp, bccomp);
}

int halo_physbc(rh_ctx *c, const MFV &m, const rh_array4 &msk, int icomp, int ncomp, int bccomp, int n_not_fill,
                const rh_array4 *calc, const rh_array4 *msku, const rh_array4 *mskv)
{
    BcScope bcs(c);                                       // all components' passes of one call share the launches
    const BcAux aux = {calc, msku, mskv};
    RH_TRY(physbc_impl(c, m, msk, icomp, ncomp, bccomp, n_not_fill, aux));
    return bcs.end();
}

// REMORA::FillPatch at level 0 (BoundaryConditions/REMORA_FillPatch.cpp:96-129) == FillBoundary + physbc
int halo_fillpatch(rh_ctx *c, const MFV &m, const rh_array4 &msk, int bccomp, int icomp, int ncomp,
                   const rh_array4 *calc, const rh_array4 *msku, const rh_array4 *mskv)
{
    RH_TRY(halo_fill_boundary(c, m, icomp, ncomp));
    return halo_physbc(c, m, msk, icomp, ncomp, bccomp, 0, calc, msku, mskv);
}

// ---- C ABI -----------------------------------------------------------------------------------------------------
static MFV mfv_from_ng(const rh_ctx *c, const rh_array4 &a, const int ng[3])
{
    // index type follows from the array extent: valid = box shrunk by ng, compared with the rank's cell box
    MFV m; m.a = a;
    for (int d = 0; d < 3; d++) { m.valid.lo[d] = a.lo[d] + ng[d]; m.valid.hi[d] = a.hi[d] - ng[d]; m.ng[d] = ng[d]; }
    for (int d = 0; d < 3; d++) m.ix[d] = (m.valid.hi[d] - m.valid.lo[d]) == (c->bx.hi[d] - c->bx.lo[d] + 1) ? 1 : 0;
    if (a.lo[2] == a.hi[2]) m.ix[2] = 0;
    return m;
}
extern "C" int rh_fill_boundary(rh_ctx *c, rh_array4 a, const int ng[3], int icomp, int ncomp)
{ TScope t(c, "fill_boundary"); return halo_fill_boundary(c, mfv_from_ng(c, a, ng), icomp, ncomp); }
extern "C" int rh_physbc(rh_ctx *c, rh_array4 a, const int ng[3], rh_array4 msk, int icomp, int ncomp, int bccomp)
{ TScope t(c, "physbc"); return halo_physbc(c, mfv_from_ng(c, a, ng), msk, icomp, ncomp, bccomp, 0); }
extern "C" int rh_fill_patch(rh_ctx *c, rh_array4 a, const int ng[3], rh_array4 msk, int bccomp, int icomp, int ncomp)
{ TScope t(c, "fill_patch"); return halo_fillpatch(c, mfv_from_ng(c, a, ng), msk, bccomp, icomp, ncomp); }
extern "C" int rh_physbc_calc(rh_ctx *c, rh_array4 a, const int ng[3], rh_array4 msk, int icomp, int ncomp, int bccomp, int n_not_fill,
                              rh_array4 calc, rh_array4 msku, rh_array4 mskv)
{
    TScope t(c, "physbc");
    return halo_physbc(c, mfv_from_ng(c, a, ng), msk, icomp, ncomp, bccomp, n_not_fill, calc.p ? &calc : nullptr, msku.p ? &msku : nullptr, mskv.p ? &mskv : nullptr);
}
extern "C" int rh_fill_patch_calc(rh_ctx *c, rh_array4 a, const int ng[3], rh_array4 msk, int bccomp, int icomp, int ncomp,
                                  rh_array4 calc, rh_array4 msku, rh_array4 mskv)
{
    TScope t(c, "fill_patch");
    return halo_fillpatch(c, mfv_from_ng(c, a, ng), msk, bccomp, icomp, ncomp, calc.p ? &calc : nullptr, msku.p ? &msku : nullptr, mskv.p ? &mskv : nullptr);
}
