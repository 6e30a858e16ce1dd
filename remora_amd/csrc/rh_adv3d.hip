// rh_adv3d.hip -- device kernels of the corrector half of the baroclinic step (REMORA::advance_3d and its leaves).
// Same mapping as rh_setup.hip: one thread per (i,j) column, i along the wavefront, k marched in registers.
#include "rh_common.h"
#include "rh_tile.h"

// tracer flux helpers (defined in rh_setup.hip; repeated here as the kernels are compiled per translation unit)
__device__ __forceinline__ double a3_tflux_u3(double t2, double t1, double t0, double tp, double m1, double m0, double mp, double H, double sgn)
{
    const double g1 = (t1 - t2) * m1, g0 = (t0 - t1) * m0, gp = (tp - t0) * mp;
    const double curv0 = -g0 + gp, curv1 = -g1 + g0;
    const double maxH = dmax_(H, 0.0), minH = dmin_(H, 0.0);
    const double a = H * 0.5 * (t0 + t1), b = (1.0 / 6.0) * (curv0 * minH + curv1 * maxH);
    return sgn < 0.0 ? a - b : a + b;
}
__device__ __forceinline__ double a3_tflux_c4(double t2, double t1, double t0, double tp, double m1, double m0, double mp, double H, int mode)
{
    const double g1 = (t1 - t2) * m1, g0 = (t0 - t1) * m0, gp = (tp - t0) * mp;
    const double grad0 = 0.5 * (g0 + gp), grad1 = 0.5 * (g1 + g0);
    const double cffb = 1.0 / 3.0;
    return mode == 0 ? H * 0.5 * (t0 + t1 - cffb * (grad0 - grad1)) : H * 0.5 * (t0 + t1) + cffb * (grad0 + grad1);
}
__device__ __forceinline__ double a3_tflux_w(const A4 &t, int i, int j, int kk, int N, double w)
{
    const double c1 = 0.5, c2 = 7.0 / 12.0, c3 = 1.0 / 12.0;
    if (kk >= 2 && kk <= N - 1) return (c2 * (t(i, j, kk - 1) + t(i, j, kk)) - c3 * (t(i, j, kk - 2) + t(i, j, kk + 1))) * w;
    if (kk == N + 1) return 0.0;
    if (kk == N) return (c2 * t(i, j, N - 1) + c1 * t(i, j, N) - c3 * t(i, j, N - 2)) * w;
    if (kk == 1) return (c2 * t(i, j, 1) + c1 * t(i, j, 0) - c3 * t(i, j, 2)) * w;
    return 0.0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::stretch_transform, Utils/REMORA_DepthStretchTransform.H:12-249 (Vtransform 2): z_w, z_r, Hz and h comp 1 from h
// and Zt_avg1.  The stretching curves sc/Cs depend on k alone and come from the table built in rh_create; the
// reference evaluates cosh/exp at every (i,j,k).  sc = [sc_w | Cs_w | sc_r | Cs_r], each nz+1 long.
// No level depends on another: one thread per (i, j) and ST_KC levels, grid.z walks k, so every block streams a few planes of each
// output (a column per thread walking all of k wrote three 34-MB-strided streams per thread).  Level k forms z_w(k), z_r(k) and
// Hz(k) = z_w(k+1) - z_w(k); z_w(k+1) at the top of a thread's range is evaluated again by the thread that owns it, with the same
// expression on the same operands, hence the same bits as the reference's difference of the two stored values.  Level nz only has z_w.
// ST_KC levels per thread.  One level per thread was latency-bound, not bandwidth-bound (PMC: 5.9 GB moved in 1.61 ms): a wave waited
// for its two 2-D loads and for a chain of scalar table loads before its first store, and 32 resident waves per CU with 1.5 KB of
// stores each cannot cover that.  Here a wave stores 12 KB, and the four stretching curves of the block's levels are staged in LDS by
// one vector load per table entry instead of 4 x (ST_KC + 1) dependent scalar loads.
#define ST_KC 8
__global__ void __launch_bounds__(RH_BX * RH_BY)
k_stretch(B3 b, A4 z_w, A4 z_r, A4 Hz, A4 h, A4 Zt, const double *__restrict__ sc, int nz, double hc)
{
    __shared__ double tab[4 * (ST_KC + 1)];                // sc_w, Cs_w, sc_r, Cs_r of levels k0 .. k0 + ST_KC
    const RhTile rh_t_ = rh_tile();
    const int i = b.lo0 + rh_t_.x * RH_BX + (int)threadIdx.x, j = b.lo1 + rh_t_.y * RH_BY + (int)threadIdx.y;
    const int k0 = rh_t_.z * ST_KC;                        // levels k0 .. k0 + ST_KC - 1 of 0 .. nz
    const int tid = (int)threadIdx.y * RH_BX + (int)threadIdx.x;
    if (tid < 4 * (ST_KC + 1)) {
        const int t = tid / (ST_KC + 1), q = tid - t * (ST_KC + 1), k = k0 + q;
        tab[tid] = sc[t * (nz + 1) + (k <= nz ? k : nz)];
    }
    const bool in = inb2(b, i, j);
    double hwater = 1.0, zt = 0.0;
    if (in) { hwater = h(i, j, 0, 0); zt = Zt(i, j, 0); }
    __syncthreads();
    if (!in) return;
    const double *sc_w = tab, *Cs_w = tab + (ST_KC + 1), *sc_r = tab + 2 * (ST_KC + 1), *Cs_r = tab + 3 * (ST_KC + 1);
    const double hinv = 1.0 / (hc + hwater);
    auto zw_at = [&](int q) -> double {                   // z_w of level k0 + q, :63-188 (the same expression for every level: same bits as stretch_zw)
        if (k0 + q == nz) return zt;
        const double cff_w = hc * sc_w[q], cff1_w = Cs_w[q];
        const double cff2_w = (cff_w + cff1_w * hwater) * hinv;
        return zt + (zt + hwater) * cff2_w;
    };
    double zw = zw_at(0);
#pragma unroll
    for (int q = 0; q < ST_KC; q++) {
        const int k = k0 + q;
        if (k > nz) break;
        if (k == 0) h(i, j, 0, 1) = zw;
        z_w(i, j, k) = zw;
        if (k != nz) {
            const double cff_r = hc * sc_r[q], cff1_r = Cs_r[q];
            const double cff2_r = (cff_r + cff1_r * hwater) * hinv;
            z_r(i, j, k) = zt + (zt + hwater) * cff2_r;
            const double zw_up = zw_at(q + 1);
            Hz(i, j, k) = zw_up - zw;
            zw = zw_up;
        }
    }
}
// The same when all five arrays are dense over exactly the box (the native state, and REMORA's MultiFabs with their 3 ghost cells): a
// column is an index into the plane, so the launch is linear over the plane (256 columns per block, whatever the row length), the level
// chunk is blockIdx.y, and the stretching curves are scalar loads (k is uniform over the block).  tools/ubench/stores.hip measures
// what this access pattern stores at (profiles/r3_ubench_stores.txt: 1.09 ms for the three arrays; the tiled form above took 1.57).
__global__ void __launch_bounds__(256)
k_stretch_flat(double *__restrict__ z_w, double *__restrict__ z_r, double *__restrict__ Hz, double *__restrict__ h, const double *__restrict__ Zt,
               const double *__restrict__ sc, int nxy, int nz, double hc)
{
    const int col = blockIdx.x * 256 + (int)threadIdx.x;
    if (col >= nxy) return;
    const int k0 = blockIdx.y * ST_KC;
    const double hwater = h[col], zt = Zt[col];
    const double *sc_w = sc, *Cs_w = sc + (nz + 1), *sc_r = sc + 2 * (nz + 1), *Cs_r = sc + 3 * (nz + 1);
    const double hinv = 1.0 / (hc + hwater);
    auto zw_at = [&](int k) -> double {
        if (k == nz) return zt;
        const double cff_w = hc * sc_w[k], cff1_w = Cs_w[k];
        const double cff2_w = (cff_w + cff1_w * hwater) * hinv;
        return zt + (zt + hwater) * cff2_w;
    };
    double zw = zw_at(k0);
#pragma unroll
    for (int q = 0; q < ST_KC; q++) {
        const int k = k0 + q;
        if (k > nz) break;
        const size_t o = (size_t)k * nxy + col;
        if (k == 0) h[(size_t)nxy + col] = zw;
        z_w[o] = zw;
        if (k != nz) {
            const double cff_r = hc * sc_r[k], cff1_r = Cs_r[k];
            const double cff2_r = (cff_r + cff1_r * hwater) * hinv;
            z_r[o] = zt + (zt + hwater) * cff2_r;
            const double zw_up = zw_at(k + 1);
            Hz[o] = zw_up - zw;
            zw = zw_up;
        }
    }
}
extern "C" int rh_stretch_transform(rh_ctx *c, rh_state *s)
{
    TScope t(c, "stretch_transform");
    const int nz = c->nz;
    rh_box b = bx_grow(c->bx, 3, 3, 0); b.lo[2] = 0; b.hi[2] = nz - 1;
    rh_box bw = b; bw.hi[2] = nz;
    const rh_box b2 = bx_slab(b, 2, 0);
    RH_NEED(s->z_w, bw); RH_NEED(s->z_r, b); RH_NEED(s->Hz, b); RH_NEED(s->h, b2); RH_NEED(s->Zt_avg1, b2);
    if (s->h.ncomp < 2) return rh_fail("rh_stretch_transform: h needs 2 components");
    const double hc = -std::min(c->prm.prob_hi[2], -c->prm.tcline);                          // :51
    auto dense_on = [&](const rh_array4 &a, int klo, int khi) {
        return a.lo[0] == b.lo[0] && a.hi[0] == b.hi[0] && a.lo[1] == b.lo[1] && a.hi[1] == b.hi[1] && a.lo[2] == klo && a.hi[2] == khi; };
    const long nxy = (long)(b.hi[0] - b.lo[0] + 1) * (b.hi[1] - b.lo[1] + 1);
    if (!c->unfused && !c->no_flatstretch && dense_on(s->z_w, 0, nz) && dense_on(s->z_r, 0, nz - 1) && dense_on(s->Hz, 0, nz - 1) && dense_on(s->h, 0, 0) &&
        dense_on(s->Zt_avg1, 0, 0) && nxy * (nz + 1) < (1L << 31)) {
        const dim3 gf((unsigned)((nxy + 255) / 256), (unsigned)((nz + 1 + ST_KC - 1) / ST_KC), 1);
        k_stretch_flat<<<gf, 256, 0, c->stream>>>(s->z_w.p, s->z_r.p, s->Hz.p, s->h.p, s->Zt_avg1.p, c->d_sc, (int)nxy, nz, hc);
    } else {
        dim3 gs = grid2(b); gs.z = (unsigned)((nz + 1 + ST_KC - 1) / ST_KC);
        k_stretch<<<gs, rh_block(), 0, c->stream>>>(mkB(b), mkA(s->z_w), mkA(s->z_r), mkA(s->Hz), mkA(s->h), mkA(s->Zt_avg1), c->d_sc, nz, hc);
    }
    RH_LAUNCH_CHECK();
    BcScope xs(c);                                                                           // one message per neighbour for the three
    RH_TRY(halo_fill_boundary(c, mfv(c, s->z_w, 0, 0, 1), 0, 1));                            // :198-201
    RH_TRY(halo_fill_boundary(c, mfv(c, s->z_r, 0, 0, 0), 0, 1));
    RH_TRY(halo_fill_boundary(c, mfv(c, s->Hz, 0, 0, 0), 0, 1));
    return xs.end();
}

// ------------------------------------------------------------------------------------------------------------------
// u, v Adams-Bashforth update with the 3-D right-hand side, REMORA_advance_3d.cpp:104-123
__global__ void k_uv_ab3(B3 xb, B3 yb, B3 un, A4 u, A4 v, A4 ru, A4 rv, A4 Hz, A4 pm, A4 pn, int nrhs, double cff)
{
    RH_IJ(un);
    const int k = un.lo2 + RH_KZ;
    if (inb2(xb, i, j)) {
        double x = u(i, j, k);
        x += cff * (pm(i, j, 0) + pm(i - 1, j, 0)) * (pn(i, j, 0) + pn(i - 1, j, 0)) * ru(i, j, k, nrhs);
        x *= 2.0 / (Hz(i - 1, j, k) + Hz(i, j, k));
        u(i, j, k) = x;
    }
    if (inb2(yb, i, j)) {
        double x = v(i, j, k);
        x += cff * (pm(i, j, 0) + pm(i, j - 1, 0)) * (pn(i, j, 0) + pn(i, j - 1, 0)) * rv(i, j, k, nrhs);
        x *= 2.0 / (Hz(i, j - 1, k) + Hz(i, j, k));
        v(i, j, k) = x;
    }
}
int launch_uv_ab3(rh_ctx *c, rh_state *s, int istep)
{
    TScope t(c, "uv_ab3");
    const rh_box xbx = bx_nodal(c->bx, 0), ybx = bx_nodal(c->bx, 1), un = bx_union(xbx, ybx);
    const double dt = c->prm.dt;
    const int iic = istep, ntfirst = 0;
    double cff;
    if (iic == ntfirst) cff = 0.25 * dt; else if (iic == ntfirst + 1) cff = 0.25 * dt * 3.0 / 2.0; else cff = 0.25 * dt * 23.0 / 12.0;
    RH_NEED(s->xvel_new, xbx); RH_NEED(s->yvel_new, ybx); RH_NEED(s->ru, xbx); RH_NEED(s->rv, ybx);
    RH_NEED(s->Hz, bx_growlo_i(xbx)); RH_NEED(s->Hz, bx_growlo_j(ybx));
    RH_NEED(s->pm, bx_slab(bx_growlo_i(xbx), 2, 0)); RH_NEED(s->pn, bx_slab(bx_growlo_i(xbx), 2, 0));
    RH_NEED(s->pm, bx_slab(bx_growlo_j(ybx), 2, 0)); RH_NEED(s->pn, bx_slab(bx_growlo_j(ybx), 2, 0));
    k_uv_ab3<<<grid3(un), rh_block(), 0, c->stream>>>(mkB(xbx), mkB(ybx), mkB(un), mkA(s->xvel_new), mkA(s->yvel_new), mkA(s->ru), mkA(s->rv),
        mkA(s->Hz), mkA(s->pm), mkA(s->pn), 0, cff);
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::vert_visc_3d, REMORA_vert_visc_3d.cpp:10-107: implicit vertical viscosity / diffusion (parabolic-spline
// tridiagonal system on the k-faces 1..N), one thread per column: upward elimination, downward back-substitution
// fused with the final update.  Of the reference's six scratch arrays (Hzk, AK, BC, DC, FC, CF) only the two the
// back-substitution needs (CF, DC) are kept, in a work-space slab laid out like the field ([k][j][i], coalesced).
__global__ void k_vert_visc(B3 bx, int ioff, int joff, A4 phi, A4 Hz, A4 Akv, A4 wCF, A4 wDC, int N, double dt)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const int ii = i - ioff, jj = j - joff;
    const double sixth = 1.0 / 6.0, third = 1.0 / 3.0;
    double Hzk_m = 0.5 * (Hz(ii, jj, 0) + Hz(i, j, 0));
    double AK_m = 0.5 * (Akv(ii, jj, 0) + Akv(i, j, 0));
    double AK_0 = 0.5 * (Akv(ii, jj, 1) + Akv(i, j, 1));
    double phi_m = phi(i, j, 0);
    double CFp = 0.0, DCp = 0.0;
    for (int k = 1; k <= N; k++) {
        const double Hzk_0 = 0.5 * (Hz(ii, jj, k) + Hz(i, j, k));
        const double AK_p = 0.5 * (Akv(ii, jj, k + 1) + Akv(i, j, k + 1));
        const double phi_0 = phi(i, j, k);
        const double oHzkm1 = 1.0 / Hzk_m;
        const double oHz = 1.0 / Hzk_0;
        const double FC = sixth * Hzk_m - dt * AK_m / Hzk_m;
        double CF = sixth * Hzk_0 - dt * AK_p / Hzk_0;
        const double BC = third * (Hzk_m + Hzk_0) + dt * AK_0 * (oHzkm1 + oHz);
        const double cff = 1.0 / (BC - FC * CFp);
        CF *= cff;
        const double DC = cff * (phi_0 - phi_m - FC * DCp);
        wCF(i, j, k) = CF; wDC(i, j, k) = DC;
        CFp = CF; DCp = DC; Hzk_m = Hzk_0; AK_m = AK_0; AK_0 = AK_p; phi_m = phi_0;
    }
    // here AK_0 = AK(N+1), Hzk_m = Hzk(N)
    double DC_up = 0.0;                 // DC(k+1) of the back-substitution (before the AK scaling)
    double DCs_up = DC_up * AK_0;       // DC(N+1)*AK(N+1), :96-99
    for (int k = N; k >= 0; k--) {
        double DC, AKk;
        if (k >= 1) { DC = wDC(i, j, k); DC -= wCF(i, j, k) * DC_up; }
        else DC = 0.0;
        AKk = 0.5 * (Akv(ii, jj, k) + Akv(i, j, k));
        const double DCs = DC * AKk;
        const double Hzk = 0.5 * (Hz(ii, jj, k) + Hz(i, j, k));
        const double oHz = 1.0 / Hzk;
        const double cff = dt * oHz * (DCs_up - DCs);
        phi(i, j, k) += cff;
        DC_up = DC; DCs_up = DCs;
    }
}
// ---- column solver with the tridiagonal coefficients in LDS: vert_visc_3d alone (FULL = 0) or the whole velocity corrector
// (FULL = 1: Adams-Bashforth update, REMORA_advance_3d.cpp:104-123 + vert_visc_3d.cpp:10-107 + vert_mean_3d.cpp:10-47; the three
// leaves run on the same box with no ghost fill between them and each column only needs itself and its (i-1 | j-1) neighbour,
// so the velocity travels HBM -> registers -> LDS -> HBM once instead of three round trips).
// One wavefront per block, one column per lane.  CF(k), DC(k) of the elimination live in LDS ([k][lane], conflict-free)
// instead of the reference's six scratch arrays.  The (i-1) neighbour of a u column comes from the neighbouring lane by a
// cross-lane shuffle (only lane 0 loads it).  LDS limits a CU to ~3 such wavefronts, so HBM latency is hidden inside the wave:
// the operands of level k+D are requested while level k is eliminated (a register ring of D levels, statically indexed; the
// vector-memory counter lets the wave wait for the oldest level only).  D x loads-per-level stays below the 63 outstanding loads
// the counter can track.  Levels past the end are clamped, not branched around, so the ring is branch-free.
//   forward sweep  : [FULL: phi(k) = (u + c2 ru) * (2 / (Hz' + Hz)) on the fly] elimination coefficients CF, DC -> LDS
//   backward sweep : back-substitution fused with the update; FULL: phi(k) recomputed from the same operands (re-read through
//                    L2 / Infinity Cache) and the updated phi(k), Hzk(k) parked in the LDS slots of DC(k), CF(k), dead by then
//   FULL mean      : ascending sums over LDS in the reference's order, then the only store of the velocity.
struct ColLvl { double P, R, H, Hx, A, Ax; };
template <int IO, int JO, int FULL, int D>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))     // LDS caps occupancy anyway: give the ring all the registers it wants
k_column_solver(B3 bx, A4 phi, A4 rphi, A4 Hz, A4 Akv, A4 Davg1, A4 dxlen, A4 pm, A4 pn, A4 msk, int N, double dt, double cab)
{
    extern __shared__ double vv_lds[];
    const int lane = threadIdx.x;
    double *sCF = vv_lds + lane, *sDC = vv_lds + (size_t)(N + 2) * 64 + lane;      // s[k * 64]
    // IO = 1 (u): waves overlap by one column -- lane 0 only carries Hz / Akv of the (i-1) neighbour of lane 1 into the shuffle, so no
    // lane ever issues a one-element load.  JO = 1 (v): tiles are walked in 1 x 4 patches, so the row below (whose Hz / Akv this
    // wave reads as its j-1 operands) is streamed by a wave of the same XCD at the same time and is found in that XCD's L2.
    constexpr int SX = IO ? 63 : 64;
    const RhTile rh_t_ = JO ? rh_tile_patch<1, 4>() : rh_tile();
    const int i = bx.lo0 - IO + rh_t_.x * SX + lane, j = bx.lo1 + rh_t_.y;
    const bool act = i <= bx.hi0 && !(IO && lane == 0);
    const int ic = i <= bx.hi0 ? i : bx.hi0;               // lanes past the box shadow the last column (no stores)
    const int ip = (IO && lane == 0) ? ic + 1 : ic;        // lane 0 of a u wave: phi / ru of its own column are never used; read lane 1's
    const int jj = j - JO;
    const double sixth = 1.0 / 6.0, third = 1.0 / 3.0;
    // second point of the face average: (i-1,j) via shuffle, (i,j-1) via its own load, or the same point (tracers)
    auto second = [&](double own, double extra) -> double {
        if (IO == 1) return __shfl_up(own, 1);
        if (JO == 1) return extra;
        return own;
    };
    auto ld2 = [&](const A4 &a, int k) -> double {       // the load that `second` may need
        if (JO == 1) return a(ic, jj, k);
        return 0.0;
    };
    // k-invariant factor of the Adams-Bashforth term: cff * (pm + pm') * (pn + pn'), advance_3d.cpp:113-121
    double c2 = 0.0;
    if (FULL) c2 = cab * (pm(ip, j, 0) + pm(ip - IO, jj, 0)) * (pn(ip, j, 0) + pn(ip - IO, jj, 0));
    auto start = [&](double u, double r, double hsum) -> double {     // the field the solve starts from
        if (!FULL) return u;
        double x = u; x += c2 * r; x *= 2.0 / hsum; return x;
    };
    auto load_f = [&](int k) -> ColLvl {                 // forward operands of level k: phi, Hz at k, Akv at k+1
        const int kc = k < N ? k : N;
        ColLvl v; v.P = phi(ip, j, kc); v.R = FULL ? rphi(ip, j, kc) : 0.0; v.H = Hz(ic, j, kc); v.Hx = ld2(Hz, kc);
        v.A = Akv(ic, j, kc + 1); v.Ax = ld2(Akv, kc + 1); return v;
    };
    auto load_b = [&](int k) -> ColLvl {                 // backward operands of level k
        const int kc = k > 0 ? k : 0;
        ColLvl v; v.P = phi(ip, j, kc); v.R = FULL ? rphi(ip, j, kc) : 0.0; v.H = Hz(ic, j, kc); v.Hx = ld2(Hz, kc);
        v.A = Akv(ic, j, kc); v.Ax = ld2(Akv, kc); return v;
    };
    // ---- upward elimination, vert_visc_3d.cpp:51-75
    double Hzk_m, AK_m, AK_0, phi_m;
    {
        const double h = Hz(ic, j, 0), hx = ld2(Hz, 0), a0 = Akv(ic, j, 0), a0x = ld2(Akv, 0), a1 = Akv(ic, j, 1), a1x = ld2(Akv, 1);
        const double hs = second(h, hx) + h;
        Hzk_m = 0.5 * hs;
        AK_m = 0.5 * (second(a0, a0x) + a0);
        AK_0 = 0.5 * (second(a1, a1x) + a1);
        phi_m = start(phi(ip, j, 0), FULL ? rphi(ip, j, 0) : 0.0, hs);
    }
    double CFp = 0.0, DCp = 0.0, oHz_m = 1.0 / Hzk_m;
    auto fwd = [&](int k, const ColLvl &c) {              // level k of the elimination (all lanes: `second` shuffles)
        const double hs = second(c.H, c.Hx) + c.H;
        const double Hzk_0 = 0.5 * hs;
        const double AK_p = 0.5 * (second(c.A, c.Ax) + c.A);
        const double phi_0 = start(c.P, c.R, hs);
        const double oHzkm1 = oHz_m;                      // 1 / Hzk(k-1): the previous level's quotient, same operands, same bits
        const double oHz = 1.0 / Hzk_0;
        const double FC = sixth * Hzk_m - dt * AK_m / Hzk_m;
        double CF = sixth * Hzk_0 - dt * AK_p / Hzk_0;
        const double BC = third * (Hzk_m + Hzk_0) + dt * AK_0 * (oHzkm1 + oHz);
        const double cff = 1.0 / (BC - FC * CFp);
        CF *= cff;
        const double DC = cff * (phi_0 - phi_m - FC * DCp);
        sCF[k * 64] = CF; sDC[k * 64] = DC;
        // without the AB3 term phi_0 IS the loaded register: carried into the next level as phi_m it would outlive the ring slot,
        // and the copies that keeps alive make every loop iteration start by draining the ring.  An opaque move cuts the tie.
        double carry = phi_0; asm volatile("" : "+v"(carry));
        CFp = CF; DCp = DC; Hzk_m = Hzk_0; AK_m = AK_0; AK_0 = AK_p; phi_m = carry; oHz_m = oHz;
    };
    ColLvl ring[D];
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = load_f(1 + u);
    int kb = 1;
    for (; kb + D - 1 <= N; kb += D) {                    // full batches: straight-line code, one level's loads per level of work
#pragma unroll
        for (int u = 0; u < D; u++) {
            fwd(kb + u, ring[u]);
            __builtin_amdgcn_sched_barrier(0);             // keep the request for level k+D next to the work of level k
            ring[u] = load_f(kb + u + D);                  // into the registers level k has just released
        }
    }
#pragma unroll
    for (int u = 0; u < D; u++) if (kb + u <= N) fwd(kb + u, ring[u]);      // tail: operands are already in the ring
    // ---- back-substitution fused with the update, :81-106.  Here AK_0 = AK(N+1).
    double DC_up = 0.0, DCs_up = DC_up * AK_0;
    auto bwd = [&](int k, const ColLvl &c) {
        const double hs = second(c.H, c.Hx) + c.H;
        const double Hzk = 0.5 * hs;
        const double AKk = 0.5 * (second(c.A, c.Ax) + c.A);
        double DC;
        if (k >= 1) { DC = sDC[k * 64]; DC -= sCF[k * 64] * DC_up; }
        else DC = 0.0;
        const double DCs = DC * AKk;
        const double oHz = 1.0 / Hzk;
        const double cff = dt * oHz * (DCs_up - DCs);
        const double x = start(c.P, c.R, hs) + cff;              // phi after vert_visc_3d
        if (FULL) { sDC[k * 64] = x; sCF[k * 64] = Hzk; }
        else if (act) phi(i, j, k) = x;
        DC_up = DC; DCs_up = DCs;
    };
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = load_b(N - u);
    kb = N;
    for (; kb - D + 1 >= 1; kb -= D) {                    // full batches of levels >= 1
#pragma unroll
        for (int u = 0; u < D; u++) {
            bwd(kb - u, ring[u]);
            __builtin_amdgcn_sched_barrier(0);
            ring[u] = load_b(kb - u - D);
        }
    }
#pragma unroll
    for (int u = 0; u < D; u++) if (kb - u >= 0) bwd(kb - u, ring[u]);
    if (!FULL) return;
    // ---- vert_mean_3d: ascending sums in the reference's order (:22-34), then the correction (:36-46)
    double CF = sCF[0], DCm = sDC[0] * CF;
    for (int k = 1; k <= N; k++) { const double Hzf = sCF[k * 64]; CF += Hzf; DCm += sDC[k * 64] * Hzf; }
    const double on = 2.0 / (dxlen(ip - IO, jj, 0) + dxlen(ip, j, 0));
    const double cff1 = 1.0 / (CF * (on));
    DCm = (DCm * (on) - Davg1(ip, j, 0)) * cff1;
    const double m = msk(ip, j, 0);
    if (act)
        for (int k = 0; k <= N; k++) { double x = sDC[k * 64]; x -= DCm; x *= m; phi(i, j, k) = x; }
}
// what the velocity corrector needs to carry update_massflux_3d for its column as well (on = 0: not asked for)
struct MfTail { int on, dlox, dhix, dloy, dhiy, perx, pery; A4 Hphi, phibar, Davg2; };
// ---- the same solver with NO second read of its operands, for columns of up to VV_NZMAX levels.  The elimination coefficients
// CF(k), DC(k) and the face-averaged AK(k) stay in REGISTERS (the loops over k are fully unrolled, so every index is a compile-time
// constant; one wavefront per SIMD owns the whole 512-entry register file), the field the solve starts from and Hzk(k) are parked
// in LDS ([k][lane]).  The back-substitution then touches no global memory: each operand travels HBM -> CU exactly once
// (k_column_solver re-reads 3..6 arrays on the way down: 1.8-2.7x the algorithmic bytes, profiles/r1_v5_traffic.json).
// Same expressions in the same order as k_column_solver: bit-identical.
// The number of levels is a template parameter: with a run-time trip count every unrolled level becomes a guarded block, and the
// wait counts hipcc then places at the joins drain the load ring (measured: 1.7x slower than the ring kernel).  Instantiated for
// the depths of the pinned cases and of the roofline configuration; any other depth takes k_column_solver.
// CK = 1: the mixing coefficient is ONE number, akc (RH_VMIX_CONST once the arrays hold the constants, see mkA_vmix): no load of it on
// either sweep, and FC(k) = sixth Hzk(k-1) - dt AK / Hzk(k-1) is the value CF(k-1) had before its scaling -- the same expression on the
// same operands, hence the same bits -- so one of the four divisions per level goes.  0.5 (akc + akc) stands where the face average
// of two equal coefficients stood (exact).
template <int IO, int JO, int FULL, int D, int NZ, int CK = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_column_solver_reg(B3 bx, A4 phi, A4 rphi, A4 Hz, A4 Akv, A4 Davg1, A4 dxlen, A4 pm, A4 pn, A4 msk, int /*N*/, double dt, double cab,
                    MfTail mf, double akc)
{
    constexpr int N = NZ - 1, VV_NZMAX = NZ;
    const double akh = 0.5 * (akc + akc);
    // ONE array in LDS (the field the solve starts from / ends with, [k][lane]); cf, dc and the face thickness Hzk (tracers: its
    // reciprocal, the same quotient the way up formed) in registers; the
    // face-averaged mixing coefficient is read again on the way down (one more operand per level, in a ring of its own -- a constant
    // row under RH_VMIX_CONST).  With Hzk in LDS as well (round 2) a wave needed 2 x NZ x 512 B = 51 KB at 50 levels, 65 KB at 64:
    // three, respectively two, of the CU's four SIMDs had a wave (each wave owns its SIMD's whole register file); now all four do.
    extern __shared__ double vv_lds[];
    const int lane = threadIdx.x;
    double *sX = vv_lds + lane;                                                     // s[k * 64], k = 0 .. N
    constexpr int SX = IO ? 63 : 64;
    const RhTile rh_t_ = JO ? rh_tile_patch<1, 4>() : rh_tile();
    const int i = bx.lo0 - IO + rh_t_.x * SX + lane, j = bx.lo1 + rh_t_.y;
    const bool act = i <= bx.hi0 && !(IO && lane == 0);
    const int ic = i <= bx.hi0 ? i : bx.hi0;
    const int ip = (IO && lane == 0) ? ic + 1 : ic;
    const int jj = j - JO;
    const double sixth = 1.0 / 6.0, third = 1.0 / 3.0;
    auto second = [&](double own, double extra) -> double {
        if (IO == 1) return __shfl_up(own, 1);
        if (JO == 1) return extra;
        return own;
    };
    auto ld2 = [&](const A4 &a, int k) -> double { if (JO == 1) return a(ic, jj, k); return 0.0; };
    double c2 = 0.0;
    if (FULL) c2 = cab * (pm(ip, j, 0) + pm(ip - IO, jj, 0)) * (pn(ip, j, 0) + pn(ip - IO, jj, 0));
    // 2 / hsum of the reference is the reciprocal of the face thickness 0.5 hsum that the elimination needs anyway: halving is exact, and
    // a correctly rounded quotient depends on the real value of its operands' ratio alone, so 1 / (0.5 hsum) has the bits of 2 / hsum
    auto start = [&](double u, double r, double ohz) -> double {
        if (!FULL) return u;
        double x = u; x += c2 * r; x *= ohz; return x;
    };
    auto load_f = [&](int k) -> ColLvl {
        const int kc = k < N ? k : N;
        ColLvl v; v.P = phi(ip, j, kc); v.R = FULL ? rphi(ip, j, kc) : 0.0; v.H = Hz(ic, j, kc); v.Hx = ld2(Hz, kc);
        if (CK) { v.A = 0.0; v.Ax = 0.0; } else { v.A = Akv(ic, j, kc + 1); v.Ax = ld2(Akv, kc + 1); }
        return v;
    };
    // deeper than 50 levels the three per-level register arrays no longer fit beside the load rings (22-58 registers spilled at 64): the
    // thickness of the levels from KS up goes to LDS behind sX (40 KB a wave at 64 levels: four waves still share a CU)
    constexpr int KS = NZ > 50 ? 50 : NZ;
    double *sH = vv_lds + NZ * 64 + lane;
    double cf[VV_NZMAX], dc[VV_NZMAX], hzk[KS];
#define HZ_GET(k) ((k) < KS ? hzk[(k) < KS ? (k) : 0] : sH[((k) - KS) * 64])
#define HZ_SET(k, v) do { if ((k) < KS) hzk[(k) < KS ? (k) : 0] = (v); else sH[((k) - KS) * 64] = (v); } while (0)
    double Hzk_m, AK_m, AK_0, phi_m;
    {
        const double h = Hz(ic, j, 0), hx = ld2(Hz, 0);
        const double hs = second(h, hx) + h;
        Hzk_m = 0.5 * hs;
        if (CK) { AK_m = akh; AK_0 = akh; }
        else {
            const double a0 = Akv(ic, j, 0), a0x = ld2(Akv, 0), a1 = Akv(ic, j, 1), a1x = ld2(Akv, 1);
            AK_m = 0.5 * (second(a0, a0x) + a0);
            AK_0 = 0.5 * (second(a1, a1x) + a1);
        }
        phi_m = start(phi(ip, j, 0), FULL ? rphi(ip, j, 0) : 0.0, 1.0 / Hzk_m);
    }
    double CFp = 0.0, DCp = 0.0, oHz_m = 1.0 / Hzk_m;
    double CFraw_m = CK ? sixth * Hzk_m - dt * AK_m / Hzk_m : 0.0;       // CK: what FC(1) is
    cf[0] = 0.0; dc[0] = 0.0; hzk[0] = FULL ? Hzk_m : oHz_m;
    static_assert(KS >= 1, "level 0 stays in a register");
    sX[0] = phi_m;
    ColLvl ring[D];
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = load_f(1 + u);
    // ---- upward elimination, vert_visc_3d.cpp:51-75
#pragma clang loop unroll(full)
    for (int k = 1; k < VV_NZMAX; k++) {
        {
        const ColLvl c = ring[(k - 1) % D];
        const double hs = second(c.H, c.Hx) + c.H;
        const double Hzk_0 = 0.5 * hs;
        const double AK_p = CK ? akh : 0.5 * (second(c.A, c.Ax) + c.A);
        const double oHzkm1 = oHz_m;
        const double oHz = 1.0 / Hzk_0;
        const double phi_0 = start(c.P, c.R, oHz);
        const double FC = CK ? CFraw_m : sixth * Hzk_m - dt * AK_m / Hzk_m;
        double CF = sixth * Hzk_0 - dt * AK_p / Hzk_0;
        CFraw_m = CF;
        const double BC = third * (Hzk_m + Hzk_0) + dt * AK_0 * (oHzkm1 + oHz);
        const double cff = 1.0 / (BC - FC * CFp);
        CF *= cff;
        const double DC = cff * (phi_0 - phi_m - FC * DCp);
        cf[k] = CF; dc[k] = DC; HZ_SET(k, FULL ? Hzk_0 : oHz);      // tracers only ever need the reciprocal again
        sX[k * 64] = phi_0;
        __builtin_amdgcn_sched_barrier(0);
        ring[(k - 1) % D] = load_f(k + D);
        double carry = phi_0; asm volatile("" : "+v"(carry));
        CFp = CF; DCp = DC; Hzk_m = Hzk_0; AK_m = AK_0; AK_0 = AK_p; phi_m = carry; oHz_m = oHz;
        }
    }
    // ---- back-substitution fused with the update, :81-106.  Here AK_0 = AK(N+1).  The only global loads are the mixing
    // coefficients of the level (the same operands the way up averaged, through the same expression: the same AK(k)), DB levels ahead.
    constexpr int DB = NZ > 50 ? 4 : 12;
    struct AkL { double A, Ax; };
    auto load_a = [&](int k) -> AkL { const int kc = k > 0 ? k : 0; AkL v; if (CK) { v.A = 0.0; v.Ax = 0.0; } else { v.A = Akv(ic, j, kc); v.Ax = ld2(Akv, kc); } return v; };
    AkL aring[DB];
#pragma unroll
    for (int u = 0; u < DB; u++) aring[u] = load_a(N - u);
    double DC_up = 0.0, DCs_up = DC_up * AK_0;
#pragma clang loop unroll(full)
    for (int k = VV_NZMAX - 1; k >= 0; k--) {
        {
            const AkL a = aring[(N - k) % DB];
            const double AKk = CK ? akh : 0.5 * (second(a.A, a.Ax) + a.A);
            double DC;
            if (k >= 1) { DC = dc[k]; DC -= cf[k] * DC_up; }
            else DC = 0.0;
            const double DCs = DC * AKk;
            // FULL: the thickness itself is kept (the depth mean needs it) and its reciprocal is formed again here -- through an opaque
            // move, or hipcc recognises the quotient of the way up and keeps all NZ of them alive instead (100-128 more registers)
            double Hzk = HZ_GET(k);
            if (FULL) asm volatile("" : "+v"(Hzk));
            const double oHz = FULL ? 1.0 / Hzk : Hzk;
            const double cff = dt * oHz * (DCs_up - DCs);
            const double x = sX[k * 64] + cff;
            if (FULL) sX[k * 64] = x;
            else if (act) phi(i, j, k) = x;
            __builtin_amdgcn_sched_barrier(0);
            aring[(N - k) % DB] = load_a(k - DB);
            DC_up = DC; DCs_up = DCs;
        }
    }
    if (!FULL) return;
    // the old mass flux of the fused update_massflux_3d tail is requested here, while the depth mean is being summed
    constexpr int MD = NZ > 50 ? 6 : 20;
    double hq[MD];
    if (mf.on) {
#pragma unroll
        for (int u = 0; u < MD; u++) hq[u] = mf.Hphi(ip, j, u < N ? u : N);
    }
    // ---- vert_mean_3d: ascending sums in the reference's order (:22-34), then the correction (:36-46)
    double CF = hzk[0], DCm = sX[0] * CF;
#pragma clang loop unroll(full)
    for (int k = 1; k <= N; k++) { const double Hzf = HZ_GET(k); CF += Hzf; DCm += sX[k * 64] * Hzf; }
    const double on = 2.0 / (dxlen(ip - IO, jj, 0) + dxlen(ip, j, 0));
    const double cff1 = 1.0 / (CF * (on));
    DCm = (DCm * (on) - Davg1(ip, j, 0)) * cff1;
    const double m = msk(ip, j, 0);
    if (!mf.on) {
        if (act)
            for (int k = 0; k <= N; k++) { double x = sX[k * 64]; x -= DCm; x *= m; phi(i, j, k) = x; }
        return;
    }
    // ---- update_massflux_3d (REMORA_update_massflux_3d.cpp:24-122) for this column while its final velocity is still in LDS and its
    // layer thickness in registers: Hphi = 1/2 (Hphi + phi DC) with its depth integral replaced by Davg2, phibar.  Same expressions and
    // the same ascending sums as k_massflux_lds; DC(k) = (0.5 on) (Hz + Hz') is rebuilt from the kept Hzk = 0.5 (Hz' + Hz) (the factor 2
    // is exact).  Columns on a wall face are left out: the reference corrects their velocity after the physical BCs have been applied
    // to it (advance_3d.cpp:154-155, then :87-101 of update_massflux_3d), so they wait for the strips pass that follows physbc.
    const bool edge = (ip == mf.dlox - JO && !mf.perx) || (ip == mf.dhix + 1 && !mf.perx) || (j == mf.dloy - IO && !mf.pery) || (j == mf.dhiy + 1 && !mf.pery);
    const bool mfa = act && !edge;
    const double hon = 0.5 * on;
    double DCs = 0.0, FCm = 0.0;
#pragma clang loop unroll(full)
    for (int k = 0; k < VV_NZMAX; k++) {
        double x = sX[k * 64]; x -= DCm; x *= m;
        if (act) phi(i, j, k) = x;
        const double DCk = hon * (2.0 * HZ_GET(k));
        DCs += DCk;
        const double hp = 0.5 * (hq[k % MD] + x * DCk);
        FCm += hp;
        sX[k * 64] = hp;
        hq[k % MD] = mf.Hphi(ip, j, k + MD < N ? k + MD : N);
    }
    const double DCi = 1.0 / DCs;
    const double a1 = Davg1(ip, j, 0);
    FCm = DCi * (FCm - mf.Davg2(ip, j, 0));
    if (mfa) {
#pragma clang loop unroll(full)
        for (int k = 0; k < VV_NZMAX; k++) mf.Hphi(i, j, k) = sX[k * 64] - (hon * (2.0 * HZ_GET(k))) * FCm;
        const double pb = DCi * a1;
        mf.phibar(i, j, 0, 0) = pb; mf.phibar(i, j, 0, 1) = pb;
    }
}
#undef HZ_GET
#undef HZ_SET
static inline bool vv_reg_depth(int nz) { return nz == 13 || nz == 16 || nz == 50 || nz == 64; }
#define VV_D_T 16      // ring depth, tracers (3 loads per level)
#define VV_D_V 10      // ring depth, v (6 loads per level)
#define VV_D_U 14      // ring depth, u (4 loads per level: the i-1 operands come from the neighbouring lane)
// ring depths of the register solver: at 64 levels (BASELINE.json configs[4]) cf / dc / ak alone take 384 of the 512 registers a lane
// has at one wave per SIMD, so the ring shrinks to what still fits without scratch spills
template <int NZ> struct VVD { static constexpr int T = NZ > 50 ? 8 : VV_D_T, V = NZ > 50 ? 5 : VV_D_V, U = NZ > 50 ? 7 : VV_D_U; };

extern "C" int rh_vert_visc_3d(rh_ctx *c, rh_box phi_bx, int ioff, int joff, rh_array4 phi, rh_array4 Hz, rh_array4 Akv, int N, double dt)
{ return vert_visc_3d_view(c, phi_bx, ioff, joff, phi, Hz, Akv, mkA(Akv), N, dt); }
// akA: what the kernels read as Akv -- the array itself (the entry point above) or the constant rows of mkA_vmix (the drivers)
int vert_visc_3d_view(rh_ctx *c, rh_box phi_bx, int ioff, int joff, rh_array4 phi, rh_array4 Hz, rh_array4 Akv, A4 akA, int N, double dt)
{
    TScope t(c, "vert_visc_3d");
    if ((ioff != 0 && ioff != 1) || (joff != 0 && joff != 1) || (ioff && joff)) return rh_fail("rh_vert_visc_3d: bad (ioff,joff)");
    phi_bx.lo[2] = 0; phi_bx.hi[2] = N;
    rh_box bs = phi_bx; bs.lo[0] -= ioff; bs.lo[1] -= joff;
    rh_box bw = bs; bw.hi[2] = N + 1;
    rh_box pw = phi_bx; pw.hi[2] = N + 1;
    RH_NEED(phi, phi_bx); RH_NEED(Hz, bs); RH_NEED(Akv, bw); RH_NEED(c->wCF, pw); RH_NEED(c->wDC, pw);
    const size_t lds_bytes = sizeof(double) * 2 * (size_t)(N + 2) * 64;
    if (!c->unfused && lds_bytes <= RH_LDS_MAX) {        // coefficients in LDS (up to 78 levels at >= 2 wavefronts per CU)
        RH_HIP(rh_lds_optin(k_column_solver<1, 0, 0, VV_D_U>, lds_bytes)); RH_HIP(rh_lds_optin(k_column_solver<0, 1, 0, VV_D_V>, lds_bytes));
        RH_HIP(rh_lds_optin(k_column_solver<0, 0, 0, VV_D_T>, lds_bytes));
        const dim3 grid((phi_bx.hi[0] - phi_bx.lo[0] + 64) / 64, phi_bx.hi[1] - phi_bx.lo[1] + 1, 1);
        const dim3 gridu((phi_bx.hi[0] - phi_bx.lo[0] + 63) / 63, phi_bx.hi[1] - phi_bx.lo[1] + 1, 1);     // u waves overlap by one column
        const A4 z = mkA(phi);                           // unused operands of the FULL form
        MfTail mfz; memset(&mfz, 0, sizeof mfz);
        const size_t lds_reg = sizeof(double) * (size_t)(N + 1 + (N + 1 > 50 ? N + 1 - 50 : 0)) * 64;   // + the thickness of the levels from 50 up (k_column_solver_reg: KS)
        if (vv_reg_depth(N + 1) && !c->noregsolve) {     // a depth the register form is built for: no operand is read twice
            double akc = 0.0;
            const bool ck = vmix_row_value(c, akA, &akc) && !c->no_ckonst;
#define RH_VVR1(NZ, CK) do { \
            if (ioff) k_column_solver_reg<1, 0, 0, VVD<NZ>::U, NZ, CK><<<gridu, 64, lds_reg, c->stream>>>(mkB(phi_bx), mkA(phi), z, mkA(Hz), akA, z, z, z, z, z, N, dt, 0.0, mfz, akc); \
            else if (joff) k_column_solver_reg<0, 1, 0, VVD<NZ>::V, NZ, CK><<<grid, 64, lds_reg, c->stream>>>(mkB(phi_bx), mkA(phi), z, mkA(Hz), akA, z, z, z, z, z, N, dt, 0.0, mfz, akc); \
            else k_column_solver_reg<0, 0, 0, VVD<NZ>::T, NZ, CK><<<grid, 64, lds_reg, c->stream>>>(mkB(phi_bx), mkA(phi), z, mkA(Hz), akA, z, z, z, z, z, N, dt, 0.0, mfz, akc); } while (0)
#define RH_VVR(NZ) do { if (ck) RH_VVR1(NZ, 1); else RH_VVR1(NZ, 0); } while (0)
            switch (N + 1) { case 13: RH_VVR(13); break; case 16: RH_VVR(16); break; case 64: RH_VVR(64); break; default: RH_VVR(50); break; }
#undef RH_VVR
#undef RH_VVR1
        }
        else if (ioff) k_column_solver<1, 0, 0, VV_D_U><<<gridu, 64, lds_bytes, c->stream>>>(mkB(phi_bx), mkA(phi), z, mkA(Hz), akA, z, z, z, z, z, N, dt, 0.0);
        else if (joff) k_column_solver<0, 1, 0, VV_D_V><<<grid, 64, lds_bytes, c->stream>>>(mkB(phi_bx), mkA(phi), z, mkA(Hz), akA, z, z, z, z, z, N, dt, 0.0);
        else k_column_solver<0, 0, 0, VV_D_T><<<grid, 64, lds_bytes, c->stream>>>(mkB(phi_bx), mkA(phi), z, mkA(Hz), akA, z, z, z, z, z, N, dt, 0.0);
    } else                                               // deeper columns: coefficients in the global work-space slab
        k_vert_visc<<<grid2(phi_bx), rh_block(), 0, c->stream>>>(mkB(phi_bx), ioff, joff, mkA(phi), mkA(Hz), akA, mkA(c->wCF), mkA(c->wDC), N, dt);
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// AB3 + vert_visc_3d + vert_mean_3d for u and v; *applied = 0 when the fused form does not apply (the caller runs the leaves)
int launch_uv_corrector(rh_ctx *c, rh_state *s, int istep, int *applied)
{
    const int N = c->N; const double dt = c->prm.dt;
    const size_t lds_bytes = sizeof(double) * 2 * (size_t)(N + 2) * 64;
    *applied = 0;
    if (c->unfused || lds_bytes > RH_LDS_MAX) return 0;
    RH_HIP(rh_lds_optin(k_column_solver<1, 0, 1, VV_D_U>, lds_bytes)); RH_HIP(rh_lds_optin(k_column_solver<0, 1, 1, VV_D_V>, lds_bytes));
    const int iic = istep, ntfirst = 0;
    double cab;
    if (iic == ntfirst) cab = 0.25 * dt; else if (iic == ntfirst + 1) cab = 0.25 * dt * 3.0 / 2.0; else cab = 0.25 * dt * 23.0 / 12.0;
    rh_box xbx = bx_nodal(c->bx, 0), ybx = bx_nodal(c->bx, 1);
    xbx.lo[2] = ybx.lo[2] = 0; xbx.hi[2] = ybx.hi[2] = N;
    rh_box xw = xbx, yw = ybx; xw.hi[2] = yw.hi[2] = N + 1;
    const rh_box xs = bx_growlo_i(xbx), ys = bx_growlo_j(ybx), xsw = bx_growlo_i(xw), ysw = bx_growlo_j(yw);
    RH_NEED(s->xvel_new, xbx); RH_NEED(s->yvel_new, ybx); RH_NEED(s->ru, xbx); RH_NEED(s->rv, ybx);
    RH_NEED(s->Hz, xs); RH_NEED(s->Hz, ys); RH_NEED(s->Akv, xsw); RH_NEED(s->Akv, ysw);
    RH_NEED(s->pm, bx_slab(xs, 2, 0)); RH_NEED(s->pn, bx_slab(xs, 2, 0)); RH_NEED(s->pm, bx_slab(ys, 2, 0)); RH_NEED(s->pn, bx_slab(ys, 2, 0));
    RH_NEED(s->DU_avg1, bx_slab(xbx, 2, 0)); RH_NEED(s->DV_avg1, bx_slab(ybx, 2, 0)); RH_NEED(s->msku, bx_slab(xbx, 2, 0)); RH_NEED(s->mskv, bx_slab(ybx, 2, 0));
    const dim3 gx((xbx.hi[0] - xbx.lo[0] + 63) / 63, xbx.hi[1] - xbx.lo[1] + 1, 1), gy((ybx.hi[0] - ybx.lo[0] + 64) / 64, ybx.hi[1] - ybx.lo[1] + 1, 1);
    const bool reg = vv_reg_depth(N + 1) && !c->noregsolve;
    const size_t lds_reg = sizeof(double) * (size_t)(N + 1 + (N + 1 > 50 ? N + 1 - 50 : 0)) * 64;   // + the thickness of the levels from 50 up (k_column_solver_reg: KS)
    // update_massflux_3d rides on the same sweep (the column's final velocity and thickness are in LDS when the corrector ends):
    // *applied = 2 tells the caller to run update_massflux_3d only on what this launch leaves out (ghost faces, wall faces)
    MfTail mfu, mfv; memset(&mfu, 0, sizeof mfu); memset(&mfv, 0, sizeof mfv);
    const A4 akvA = mkA_vmix(c, s, 0);
    const bool mf = reg && !c->no_mffuse;
    if (mf) {
        RH_NEED(s->Huon, xbx); RH_NEED(s->Hvom, ybx); RH_NEED(s->ubar, bx_slab(xbx, 2, 0)); RH_NEED(s->vbar, bx_slab(ybx, 2, 0));
        RH_NEED(s->DU_avg2, bx_slab(xbx, 2, 0)); RH_NEED(s->DV_avg2, bx_slab(ybx, 2, 0));
        if (s->ubar.ncomp < 2 || s->vbar.ncomp < 2) return rh_fail("launch_uv_corrector: ubar / vbar need 2 components");
        mfu.on = mfv.on = 1;
        mfu.dlox = mfv.dlox = c->dom.lo[0]; mfu.dhix = mfv.dhix = c->dom.hi[0]; mfu.dloy = mfv.dloy = c->dom.lo[1]; mfu.dhiy = mfv.dhiy = c->dom.hi[1];
        mfu.perx = mfv.perx = c->prm.periodic[0]; mfu.pery = mfv.pery = c->prm.periodic[1];
        mfu.Hphi = mkA(s->Huon); mfu.phibar = mkA(s->ubar); mfu.Davg2 = mkA(s->DU_avg2);
        mfv.Hphi = mkA(s->Hvom); mfv.phibar = mkA(s->vbar); mfv.Davg2 = mkA(s->DV_avg2);
    }
    double akc = 0.0;
    const bool ck = vmix_row_value(c, akvA, &akc) && !c->no_ckonst;
#define RH_VVU1(NZ, CK) k_column_solver_reg<1, 0, 1, VVD<NZ>::U, NZ, CK><<<gx, 64, lds_reg, c->stream>>>(mkB(xbx), mkA(s->xvel_new), mkA(s->ru), mkA(s->Hz), akvA, mkA(s->DU_avg1), mkA(s->pn), \
        mkA(s->pm), mkA(s->pn), mkA(s->msku), N, dt, cab, mfu, akc)
#define RH_VVV1(NZ, CK) k_column_solver_reg<0, 1, 1, VVD<NZ>::V, NZ, CK><<<gy, 64, lds_reg, c->stream>>>(mkB(ybx), mkA(s->yvel_new), mkA(s->rv), mkA(s->Hz), akvA, mkA(s->DV_avg1), mkA(s->pm), \
        mkA(s->pm), mkA(s->pn), mkA(s->mskv), N, dt, cab, mfv, akc)
#define RH_VVU(NZ) do { if (ck) RH_VVU1(NZ, 1); else RH_VVU1(NZ, 0); } while (0)
#define RH_VVV(NZ) do { if (ck) RH_VVV1(NZ, 1); else RH_VVV1(NZ, 0); } while (0)
    { TScope t(c, "uv_corrector");
    if (reg) { switch (N + 1) { case 13: RH_VVU(13); break; case 16: RH_VVU(16); break; case 64: RH_VVU(64); break; default: RH_VVU(50); break; } }
    else k_column_solver<1, 0, 1, VV_D_U><<<gx, 64, lds_bytes, c->stream>>>(mkB(xbx), mkA(s->xvel_new), mkA(s->ru), mkA(s->Hz), akvA, mkA(s->DU_avg1), mkA(s->pn),
        mkA(s->pm), mkA(s->pn), mkA(s->msku), N, dt, cab);
    RH_LAUNCH_CHECK(); }
    TScope t(c, "uv_corrector");
    if (reg) { switch (N + 1) { case 13: RH_VVV(13); break; case 16: RH_VVV(16); break; case 64: RH_VVV(64); break; default: RH_VVV(50); break; } }
    else k_column_solver<0, 1, 1, VV_D_V><<<gy, 64, lds_bytes, c->stream>>>(mkB(ybx), mkA(s->yvel_new), mkA(s->rv), mkA(s->Hz), akvA, mkA(s->DV_avg1), mkA(s->pm),
        mkA(s->pm), mkA(s->pn), mkA(s->mskv), N, dt, cab);
    RH_LAUNCH_CHECK();
    *applied = mf ? 2 : 1;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::vert_mean_3d, REMORA_vert_mean_3d.cpp:10-47: replace the depth mean of u (v) by the barotropic DU_avg1 (DV_avg1)
__global__ void k_vert_mean(B3 bx, int ioff, int joff, A4 phi, A4 Hz, A4 Davg1, A4 dxlen, A4 msk, int N)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const int ii = i - ioff, jj = j - joff;
    double Hzf = 0.5 * (Hz(ii, jj, 0) + Hz(i, j, 0));
    double CF = Hzf;
    double DC = phi(i, j, 0) * Hzf;
    for (int k = 1; k <= N; k++) {
        Hzf = 0.5 * (Hz(ii, jj, k) + Hz(i, j, k));
        CF += Hzf;
        DC += phi(i, j, k) * Hzf;
    }
    const double on = 2.0 / (dxlen(ii, jj, 0) + dxlen(i, j, 0));
    const double cff1 = 1.0 / (CF * (on));
    DC = (DC * (on) - Davg1(i, j, 0)) * cff1;
    const double m = msk(i, j, 0);
    for (int k = 0; k <= N; k++) {
        double x = phi(i, j, k);
        x -= DC; x *= m;
        phi(i, j, k) = x;
    }
}
extern "C" int rh_vert_mean_3d(rh_ctx *c, rh_box phi_bx, int ioff, int joff, rh_array4 phi, rh_array4 Hz, rh_array4 Dphi_avg1,
                               rh_array4 dxlen, rh_array4 msk, int N)
{
    TScope t(c, "vert_mean_3d");
    if ((ioff != 0 && ioff != 1) || (joff != 0 && joff != 1) || (ioff && joff)) return rh_fail("rh_vert_mean_3d: bad (ioff,joff)");
    phi_bx.lo[2] = 0; phi_bx.hi[2] = N;
    rh_box bs = phi_bx; bs.lo[0] -= ioff; bs.lo[1] -= joff;
    const rh_box b2 = bx_slab(phi_bx, 2, 0);
    RH_NEED(phi, phi_bx); RH_NEED(Hz, bs); RH_NEED(Dphi_avg1, b2); RH_NEED(dxlen, bx_slab(bs, 2, 0)); RH_NEED(msk, b2);
    k_vert_mean<<<grid2(phi_bx), rh_block(), 0, c->stream>>>(mkB(phi_bx), ioff, joff, mkA(phi), mkA(Hz), mkA(Dphi_avg1), mkA(dxlen), mkA(msk), N);
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::update_massflux_3d, REMORA_update_massflux_3d.cpp:24-122: Huon = 1/2 (Huon + u Hz / n) with its depth integral
// replaced by DU_avg2, boundary-face velocity correction, ubar comps 0,1.  DC(k) is recomputed instead of stored.
__global__ void k_massflux(B3 bx, int ioff, int joff, A4 phi, A4 phibar, A4 Hphi, A4 Hz, A4 pmn, A4 Davg1, A4 Davg2, A4 msk,
                           int N, int dlox, int dhix, int dloy, int dhiy, int perx, int pery)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const int ii = i - ioff, jj = j - joff;
    const double on = 2.0 / (pmn(i, j, 0) + pmn(ii, jj, 0));
    double DCs = 0.0, CF = 0.0;
    for (int k = 0; k <= N; k++) {
        const double DCk = 0.5 * on * (Hz(i, j, k) + Hz(ii, jj, k));
        DCs += DCk;
        CF += DCk * phi(i, j, k);
    }
    const double DCi = 1.0 / DCs;
    const double a1 = Davg1(i, j, 0);
    CF = DCi * (CF - a1);
    const bool edge = (i == dlox - joff && !perx) || (i == dhix + 1 && !perx) || (j == dloy - ioff && !pery) || (j == dhiy + 1 && !pery);
    if (edge) {
        const double m = msk(i, j, 0);
        for (int k = 0; k <= N; k++) { double x = phi(i, j, k); x -= CF; x *= m; phi(i, j, k) = x; }
    }
    double FC = 0.0;
    for (int k = 0; k <= N; k++) {
        const double DCk = 0.5 * on * (Hz(i, j, k) + Hz(ii, jj, k));
        const double hp = 0.5 * (Hphi(i, j, k) + phi(i, j, k) * DCk);
        Hphi(i, j, k) = hp;
        FC += hp;
    }
    FC = DCi * (FC - Davg2(i, j, 0));
    for (int k = 0; k <= N; k++) {
        const double DCk = 0.5 * on * (Hz(i, j, k) + Hz(ii, jj, k));
        Hphi(i, j, k) -= DCk * FC;
    }
    const double pb = DCi * a1;
    phibar(i, j, 0, 0) = pb; phibar(i, j, 0, 1) = pb;
}
// Same arithmetic, one wavefront per block and one column per lane, with DC(k) and the half-updated flux parked in LDS between
// the integration sweep and the correction sweep (the slab form above re-reads Hz three times and writes Hphi twice), and the
// operands of level k+D requested while level k is integrated (register ring, see k_column_solver).  Columns on a boundary face
// (`edge`, :87-101) need the depth integral before their velocity is corrected: those few lanes redo their sums from memory.
#define MF_D 10
struct MfStrips { int n; int first[5]; int gx[4]; B3 b[4]; };        // n = 0: one box, the launch grid is its tiling
template <int IO, int JO, int D>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_massflux_lds(B3 bx0, A4 phi, A4 phibar, A4 Hphi, A4 Hz, A4 pmn, A4 Davg1, A4 Davg2, A4 msk,
               int N, int dlox, int dhix, int dloy, int dhiy, int perx, int pery, MfStrips st)
{
    extern __shared__ double mf_lds[];
    const int lane = threadIdx.x;
    double *sD = mf_lds + lane, *sX = mf_lds + (size_t)(N + 1) * 64 + lane;        // s[k * 64]
    // IO = 1: waves overlap by one column, lane 0 only carries Hz of the (i-1) neighbour of lane 1 into the shuffle; JO = 1: tiles in
    // 1 x 4 patches, so the row below is streamed by a wave of the same XCD at the same time (see k_column_solver)
    constexpr int SX = IO ? 63 : 64;
    B3 bx = bx0;
    RhTile rh_t_;
    if (st.n > 0) {                                                   // several thin boxes in one launch: block -> (strip, tile), in order
        int r = (int)blockIdx.x, q = 0;
        while (q + 1 < st.n && r >= st.first[q + 1]) q++;
        r -= st.first[q];
        bx = st.b[q];
        rh_t_.x = r % st.gx[q]; rh_t_.y = r / st.gx[q];
    } else rh_t_ = JO ? rh_tile_patch<1, 4>() : rh_tile();
    const int i = bx.lo0 - IO + rh_t_.x * SX + lane, j = bx.lo1 + rh_t_.y;
    const bool act = i <= bx.hi0 && !(IO && lane == 0);
    const int ic = i <= bx.hi0 ? i : bx.hi0;
    const int ip = (IO && lane == 0) ? ic + 1 : ic;        // lane 0 of a u wave: its own column is never used; shadow lane 1's
    const int jj = j - JO;
    const double on = 2.0 / (pmn(ip, j, 0) + pmn(ip - IO, jj, 0));
    const bool edge = (ip == dlox - JO && !perx) || (ip == dhix + 1 && !perx) || (j == dloy - IO && !pery) || (j == dhiy + 1 && !pery);
    struct L { double p, f, h, hx; };
    auto load = [&](int k) -> L {
        const int kc = k < N ? k : N;
        L r; r.p = phi(ip, j, kc); r.f = Hphi(ip, j, kc); r.h = Hz(ic, j, kc);
        r.hx = (JO == 1) ? Hz(ic, jj, kc) : 0.0;
        return r;
    };
    double DCs = 0.0, CF = 0.0, FC = 0.0;
    auto step = [&](int k, const L &c) {
        double hx = c.hx;
        if (IO == 1) hx = __shfl_up(c.h, 1);
        const double DCk = 0.5 * on * (c.h + hx);                     // :44-50
        DCs += DCk;
        CF += DCk * c.p;
        const double hp = 0.5 * (c.f + c.p * DCk);                    // :104-110 (valid for columns whose velocity is not corrected)
        FC += hp;
        sD[k * 64] = DCk; sX[k * 64] = hp;
    };
    L ring[D];
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = load(u);
    int kb = 0;
    for (; kb + D - 1 <= N; kb += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            step(kb + u, ring[u]);
            __builtin_amdgcn_sched_barrier(0);
            ring[u] = load(kb + u + D);
        }
    }
#pragma unroll
    for (int u = 0; u < D; u++) if (kb + u <= N) step(kb + u, ring[u]);
    const double DCi = 1.0 / DCs;
    const double a1 = Davg1(ip, j, 0);
    CF = DCi * (CF - a1);
    if (edge) {                                                       // :87-101, then the flux sums again with the corrected velocity
        const double m = msk(ip, j, 0);
        FC = 0.0;
        for (int k = 0; k <= N; k++) {
            double x = phi(ip, j, k); x -= CF; x *= m;
            if (act) phi(i, j, k) = x;
            const double hp = 0.5 * (Hphi(ip, j, k) + x * sD[k * 64]);
            sX[k * 64] = hp; FC += hp;
        }
    }
    FC = DCi * (FC - Davg2(ip, j, 0));                                // :112-121
    if (act) {
        for (int k = 0; k <= N; k++) Hphi(i, j, k) = sX[k * 64] - sD[k * 64] * FC;
        const double pb = DCi * a1;
        phibar(i, j, 0, 0) = pb; phibar(i, j, 0, 1) = pb;
    }
}
static int massflux_box(rh_ctx *c, const char *scope, rh_box bx, int ioff, int joff, rh_array4 phi, rh_array4 phibar, rh_array4 Hphi,
                        rh_array4 Hz, rh_array4 pm_or_pn, rh_array4 Dphi_avg1, rh_array4 Dphi_avg2, rh_array4 msk, int N)
{
    TScope t(c, scope);
    if ((ioff | joff) != 1 || (ioff & joff)) return rh_fail("rh_update_massflux_3d: (ioff,joff) must be (1,0) or (0,1)");
    bx.lo[2] = 0; bx.hi[2] = N;
    rh_box bs = bx; bs.lo[0] -= ioff; bs.lo[1] -= joff;
    const rh_box b2 = bx_slab(bx, 2, 0);
    if (phibar.ncomp < 2) return rh_fail("rh_update_massflux_3d: phibar needs 2 components");
    RH_NEED(phi, bx); RH_NEED(Hphi, bx); RH_NEED(Hz, bs); RH_NEED(pm_or_pn, bx_slab(bs, 2, 0));
    RH_NEED(phibar, b2); RH_NEED(Dphi_avg1, b2); RH_NEED(Dphi_avg2, b2); RH_NEED(msk, b2);
    const size_t lds_bytes = sizeof(double) * 2 * (size_t)(N + 1) * 64;
    if (!c->unfused && lds_bytes <= RH_LDS_MAX) {
        RH_HIP(rh_lds_optin(k_massflux_lds<1, 0, MF_D>, lds_bytes)); RH_HIP(rh_lds_optin(k_massflux_lds<0, 1, MF_D>, lds_bytes));
        const int sx = ioff ? 63 : 64;                                 // u waves overlap by one column
        const dim3 grid((bx.hi[0] - bx.lo[0] + sx) / sx, bx.hi[1] - bx.lo[1] + 1, 1);
        MfStrips st0; memset(&st0, 0, sizeof st0);
#define RH_MF(IO, JO) k_massflux_lds<IO, JO, MF_D><<<grid, 64, lds_bytes, c->stream>>>(mkB(bx), mkA(phi), mkA(phibar), mkA(Hphi), mkA(Hz), mkA(pm_or_pn), \
            mkA(Dphi_avg1), mkA(Dphi_avg2), mkA(msk), N, c->dom.lo[0], c->dom.hi[0], c->dom.lo[1], c->dom.hi[1], c->prm.periodic[0], c->prm.periodic[1], st0)
        if (ioff) RH_MF(1, 0); else RH_MF(0, 1);
#undef RH_MF
    } else
        k_massflux<<<grid2(bx), rh_block(), 0, c->stream>>>(mkB(bx), ioff, joff, mkA(phi), mkA(phibar), mkA(Hphi), mkA(Hz), mkA(pm_or_pn),
            mkA(Dphi_avg1), mkA(Dphi_avg2), mkA(msk), N, c->dom.lo[0], c->dom.hi[0], c->dom.lo[1], c->dom.hi[1], c->prm.periodic[0], c->prm.periodic[1]);
    RH_LAUNCH_CHECK(); return 0;
}

extern "C" int rh_update_massflux_3d(rh_ctx *c, rh_box bx, int ioff, int joff, rh_array4 phi, rh_array4 phibar, rh_array4 Hphi,
                                     rh_array4 Hz, rh_array4 pm_or_pn, rh_array4 Dphi_avg1, rh_array4 Dphi_avg2, rh_array4 msk, int N)
{ return massflux_box(c, "update_massflux_3d", bx, ioff, joff, phi, phibar, Hphi, Hz, pm_or_pn, Dphi_avg1, Dphi_avg2, msk, N); }
// update_massflux_3d on what the velocity corrector's fused tail left out of `full` (advance_3d.cpp:213,218 run it on the cell box
// grown by two, read as a face box): the ghost faces around the valid faces `done`, and the wall faces of a non-periodic side, whose
// velocity the physical BCs have set in the meantime.  Up to four strips.
int launch_massflux_rest(rh_ctx *c, rh_box full, rh_box done, int ioff, int joff, rh_array4 phi, rh_array4 phibar, rh_array4 Hphi,
                         rh_array4 Hz, rh_array4 pm_or_pn, rh_array4 Dphi_avg1, rh_array4 Dphi_avg2, rh_array4 msk, int N)
{
    // wall faces (k_column_solver_reg: `edge`) sit on the first / last face line of the normal direction
    if (ioff) { if (!c->prm.periodic[0]) { if (done.lo[0] == c->dom.lo[0]) done.lo[0] += 1; if (done.hi[0] == c->dom.hi[0] + 1) done.hi[0] -= 1; } }
    else      { if (!c->prm.periodic[1]) { if (done.lo[1] == c->dom.lo[1]) done.lo[1] += 1; if (done.hi[1] == c->dom.hi[1] + 1) done.hi[1] -= 1; } }
    if (done.lo[0] > done.hi[0] || done.lo[1] > done.hi[1])                          // every face of this box is a wall face: nothing was done
        return massflux_box(c, "massflux_rest", full, ioff, joff, phi, phibar, Hphi, Hz, pm_or_pn, Dphi_avg1, Dphi_avg2, msk, N);
    rh_box q[4] = {full, full, full, full};
    q[0].hi[1] = done.lo[1] - 1;                                                     // rows below
    q[1].lo[1] = done.hi[1] + 1;                                                     // rows above
    q[2].lo[1] = done.lo[1]; q[2].hi[1] = done.hi[1]; q[2].hi[0] = done.lo[0] - 1;   // columns left
    q[3].lo[1] = done.lo[1]; q[3].hi[1] = done.hi[1]; q[3].lo[0] = done.hi[0] + 1;   // columns right
    const size_t lds_bytes = sizeof(double) * 2 * (size_t)(N + 1) * 64;
    if (c->unfused || lds_bytes > RH_LDS_MAX) {
        for (const rh_box &r : q) {
            if (r.lo[0] > r.hi[0] || r.lo[1] > r.hi[1]) continue;
            RH_TRY(massflux_box(c, "massflux_rest", r, ioff, joff, phi, phibar, Hphi, Hz, pm_or_pn, Dphi_avg1, Dphi_avg2, msk, N));
        }
        return 0;
    }
    // the strips as ONE launch of the column kernel (each is a few rows or columns wide: as launches of their own they cost a column
    // march's latency each)
    TScope t(c, "massflux_rest");
    MfStrips st; memset(&st, 0, sizeof st);
    const int sx = ioff ? 63 : 64;
    int total = 0;
    for (rh_box r : q) {
        if (r.lo[0] > r.hi[0] || r.lo[1] > r.hi[1]) continue;
        r.lo[2] = 0; r.hi[2] = N;
        rh_box bs = r; bs.lo[0] -= ioff; bs.lo[1] -= joff;
        const rh_box b2 = bx_slab(r, 2, 0);
        RH_NEED(phi, r); RH_NEED(Hphi, r); RH_NEED(Hz, bs); RH_NEED(pm_or_pn, bx_slab(bs, 2, 0));
        RH_NEED(phibar, b2); RH_NEED(Dphi_avg1, b2); RH_NEED(Dphi_avg2, b2); RH_NEED(msk, b2);
        const int gx = (r.hi[0] - r.lo[0] + sx) / sx, gy = r.hi[1] - r.lo[1] + 1;
        st.b[st.n] = mkB(r); st.gx[st.n] = gx; st.first[st.n] = total; total += gx * gy; st.n++;
    }
    if (!st.n) return 0;
    st.first[st.n] = total;
    if (phibar.ncomp < 2) return rh_fail("launch_massflux_rest: phibar needs 2 components");
    RH_HIP(rh_lds_optin(k_massflux_lds<1, 0, MF_D>, lds_bytes)); RH_HIP(rh_lds_optin(k_massflux_lds<0, 1, MF_D>, lds_bytes));
#define RH_MFS(IO, JO) k_massflux_lds<IO, JO, MF_D><<<total, 64, lds_bytes, c->stream>>>(st.b[0], mkA(phi), mkA(phibar), mkA(Hphi), mkA(Hz), mkA(pm_or_pn), \
        mkA(Dphi_avg1), mkA(Dphi_avg2), mkA(msk), N, c->dom.lo[0], c->dom.hi[0], c->dom.lo[1], c->dom.hi[1], c->prm.periodic[0], c->prm.periodic[1], st)
    if (ioff) RH_MFS(1, 0); else RH_MFS(0, 1);
#undef RH_MFS
    RH_LAUNCH_CHECK();
    return 0;
}
// ------------------------------------------------------------------------------------------------------------------
// REMORA::rhs_t_3d, REMORA_rhs_t_3d.cpp:27-335: corrector tracer advection with the predictor field sstore, one tracer.
// Horizontal fluxes (third-order upstream / fourth-order centred, flat-bathymetry sign quirk included), horizontal update,
// fourth-order vertical flux and division by Hz in one upward sweep; FX/FE/curv/grad/FC never touch memory.
template <int C4>
__global__ void k_rhs_t(B3 bx, A4 t, A4 ss, A4 Huon, A4 Hvom, A4 Hz, A4 pn, A4 pm, A4 W, A4 msku, A4 mskv, int N, double dt, int flat)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const double cff = dt * pm(i, j, 0) * pn(i, j, 0);
    const double mu1 = msku(i - 1, j, 0), mu0 = msku(i, j, 0), mup = msku(i + 1, j, 0), mupp = msku(i + 2, j, 0);
    const double mv1 = mskv(i, j - 1, 0), mv0 = mskv(i, j, 0), mvp = mskv(i, j + 1, 0), mvpp = mskv(i, j + 2, 0);
    const double sgn = flat ? 1.0 : -1.0;
    double FCk = 0.0;
    for (int k = 0; k <= N; k++) {
        const double s0 = ss(i, j, k);
        const double xm2 = ss(i - 2, j, k), xm1 = ss(i - 1, j, k), xp1 = ss(i + 1, j, k), xp2 = ss(i + 2, j, k);
        const double ym2 = ss(i, j - 2, k), ym1 = ss(i, j - 1, k), yp1 = ss(i, j + 1, k), yp2 = ss(i, j + 2, k);
        const double Hu0 = Huon(i, j, k), Hu1 = Huon(i + 1, j, k), Hv0 = Hvom(i, j, k), Hv1 = Hvom(i, j + 1, k);
        double FX0, FX1, FE0, FE1;
        if (!C4) {
            FX0 = a3_tflux_u3(xm2, xm1, s0, xp1, mu1, mu0, mup, Hu0, sgn); FX1 = a3_tflux_u3(xm1, s0, xp1, xp2, mu0, mup, mupp, Hu1, sgn);
            FE0 = a3_tflux_u3(ym2, ym1, s0, yp1, mv1, mv0, mvp, Hv0, sgn); FE1 = a3_tflux_u3(ym1, s0, yp1, yp2, mv0, mvp, mvpp, Hv1, sgn);
        } else {
            FX0 = a3_tflux_c4(xm2, xm1, s0, xp1, mu1, mu0, mup, Hu0, flat); FX1 = a3_tflux_c4(xm1, s0, xp1, xp2, mu0, mup, mupp, Hu1, flat);
            FE0 = a3_tflux_c4(ym2, ym1, s0, yp1, mv1, mv0, mvp, Hv0, flat); FE1 = a3_tflux_c4(ym1, s0, yp1, yp2, mv0, mvp, mvpp, Hv1, flat);
        }
        double x = t(i, j, k);
        const double cff1 = cff * (FX1 - FX0);
        const double cff2 = cff * (FE1 - FE0);
        const double cff3 = cff1 + cff2;
        x -= cff3;
        const double FCk1 = a3_tflux_w(ss, i, j, k + 1, N, W(i, j, k + 1));
        const double cff4 = FCk1 - FCk;
        t(i, j, k) = (x - cff * cff4) / Hz(i, j, k);
        FCk = FCk1;
    }
}
extern "C" int rh_rhs_t_3d(rh_ctx *c, rh_box bx, rh_array4 t, rh_array4 sstore, rh_array4 Huon, rh_array4 Hvom, rh_array4 Hz,
                           rh_array4 pn, rh_array4 pm, rh_array4 W, rh_array4 msku, rh_array4 mskv, int N, double dt)
{
    TScope ts(c, "rhs_t_3d");
    bx.lo[2] = 0; bx.hi[2] = N;
    const rh_box g2 = bx_grow(bx, 2, 2, 0), b2 = bx_slab(bx, 2, 0);
    rh_box bw = bx; bw.hi[2] = N + 1;
    RH_NEED(sstore, g2); RH_NEED(t, bx); RH_NEED(Hz, bx); RH_NEED(W, bw);
    RH_NEED(Huon, bx_nodal(bx, 0)); RH_NEED(Hvom, bx_nodal(bx, 1)); RH_NEED(pm, b2); RH_NEED(pn, b2);
    { rh_box m = b2; m.lo[0] -= 1; m.hi[0] += 2; RH_NEED(msku, m); }
    { rh_box m = b2; m.lo[1] -= 1; m.hi[1] += 2; RH_NEED(mskv, m); }
#define RH_RT(C4) k_rhs_t<C4><<<grid2(bx), rh_block(), 0, c->stream>>>(mkB(bx), mkA(t), mkA(sstore), mkA(Huon), mkA(Hvom), mkA(Hz), mkA(pn), mkA(pm), \
        mkA(W), mkA(msku), mkA(mskv), N, dt, c->prm.flat_bathymetry ? 1 : 0)
    if (c->prm.tracer_c4) RH_RT(1); else RH_RT(0);
#undef RH_RT
    RH_LAUNCH_CHECK(); return 0;
}

// LDS-tiled tracer sweep for all NCONS tracers, shared by the predictor and the corrector:
//   MODE 0 = prestep_t_advection + tracer prestep_diffusion + t3dmix (arithmetic of k_prestep_t3 in rh_setup.hip)
//   MODE 1 = rhs_t_3d                                                (arithmetic of k_rhs_t3 above)
// Per level every thread loads its own point of each field exactly once (the advected field two levels ahead, into a
// 4-level register window for the vertical flux), the halo ring is fetched by the first threads of the block, and every
// horizontal neighbour comes from LDS.  One barrier per level (tiles double-buffered on k parity).
struct TrcPar { int N, flat; double cff, cff1, cff2, dt; int mix_on[RH_NCONS], mix_comp[RH_NCONS], mix_k[RH_NCONS]; };

__device__ __forceinline__ double win_flux_w(double tm1, double t0, double tp1, double tp2, int kk, int N, double w)
{   // fourth-order vertical flux through face kk from the window t(kk-2), t(kk-1), t(kk), t(kk+1) = tm1, t0, tp1, tp2
    const double c1 = 0.5, c2 = 7.0 / 12.0, c3 = 1.0 / 12.0;
    if (kk >= 2 && kk <= N - 1) return (c2 * (t0 + tp1) - c3 * (tm1 + tp2)) * w;
    if (kk == N + 1) return 0.0;
    if (kk == N) return (c2 * t0 + c1 * tp1 - c3 * tm1) * w;         // t(N-1)=t0, t(N)=tp1, t(N-2)=tm1
    if (kk == 1) return (c2 * tp1 + c1 * t0 - c3 * tp2) * w;         // t(1)=tp1, t(0)=t0, t(2)=tp2
    return 0.0;
}

template <int MODE, int SCHEME>     // SCHEME: 0 upstream3, 1 centered4, 2 flat-bathymetry centred flux (predictor only)
__global__ void __launch_bounds__(RT_BX * RT_BY, 2)
k_tracers_tiled(B3 bx, A4 adv /* cons_old | sstore */, A4 cnew, A4 sst, A4 Hz, A4 Huon, A4 Hvom, A4 W, A4 pm, A4 pn, A4 msku, A4 mskv,
                A4 diff2, TrcPar P)
{
    typedef Tile<2, 2> TT; typedef Tile<1, 0> TU; typedef Tile<0, 1> TV; typedef Tile<1, 1> TH;
    constexpr int NT = RT_BX * RT_BY;
    constexpr int SET = RH_NCONS * TT::SIZE + TU::SIZE + TV::SIZE + (MODE == 0 ? TH::SIZE : 0);
    // k-invariant per-thread values live in LDS, not registers: 8 mask values, and (predictor) 4 mixing coefficients per tracer
    constexpr int NINV = 8 + (MODE == 0 ? 4 * RH_NCONS : 0);
    __shared__ double lds[2 * SET + NINV * NT];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * RT_BX + tx;
    double *inv = lds + 2 * SET + tid;                 // inv[q * NT]: conflict-free
    const RhTile rh_t_ = rh_tile_patch<8, 4>();          // one block per CU (LDS): 32 resident per XCD
    const int i0 = bx.lo0 + rh_t_.x * RT_BX, j0 = bx.lo1 + rh_t_.y * RT_BY;
    const int i = i0 + tx, j = j0 + ty;
    const bool act = inb2(bx, i, j);
    const int N = P.N;
    const int ic = act ? i : bx.lo0, jc = act ? j : bx.lo1;     // inactive threads: keep the (unused) own-point loads in bounds
    const double pm0 = pm(ic, jc, 0), pn0 = pn(ic, jc, 0);
    const double pmn = (MODE == 0 ? P.cff : P.dt) * pm0 * pn0;
    const double cmix = P.dt * pm0 * pn0;
    if (act) {
        inv[0 * NT] = msku(i - 1, j, 0); inv[1 * NT] = msku(i, j, 0); inv[2 * NT] = msku(i + 1, j, 0); inv[3 * NT] = msku(i + 2, j, 0);
        inv[4 * NT] = mskv(i, j - 1, 0); inv[5 * NT] = mskv(i, j, 0); inv[6 * NT] = mskv(i, j + 1, 0); inv[7 * NT] = mskv(i, j + 2, 0);
        if (MODE == 0) {
            const double pmon_u0 = (pm(i - 1, j, 0) + pm0) / (pn(i - 1, j, 0) + pn0), pmon_u1 = (pm0 + pm(i + 1, j, 0)) / (pn0 + pn(i + 1, j, 0));
            const double pnom_v0 = (pn(i, j - 1, 0) + pn0) / (pm(i, j - 1, 0) + pm0), pnom_v1 = (pn0 + pn(i, j + 1, 0)) / (pm0 + pm(i, j + 1, 0));
#pragma unroll
            for (int n = 0; n < RH_NCONS; n++) {
                double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
                if (P.mix_on[n]) {
                    const int dk = P.mix_k[n], dc = P.mix_comp[n];
                    const double d0 = diff2(i, j, dk, dc);
                    a0 = 0.25 * (d0 + diff2(i - 1, j, dk, dc)) * pmon_u0; a1 = 0.25 * (diff2(i + 1, j, dk, dc) + d0) * pmon_u1;
                    b0 = 0.25 * (d0 + diff2(i, j - 1, dk, dc)) * pnom_v0; b1 = 0.25 * (diff2(i, j + 1, dk, dc) + d0) * pnom_v1;
                }
                inv[(8 + 4 * n) * NT] = a0; inv[(9 + 4 * n) * NT] = a1; inv[(10 + 4 * n) * NT] = b0; inv[(11 + 4 * n) * NT] = b1;
            }
        }
    }
    const double sgn = (MODE == 1 && P.flat) ? 1.0 : -1.0;
    const int c4mode = (MODE == 1 && P.flat) ? 1 : 0;
    // this thread's halo cell in each tile shape (the three tracer tiles share the geometry of `adv`)
    const HaloCell ht = halo_cell<2, 2>(adv, i0, j0, tx, ty), hu = halo_cell<1, 0>(Huon, i0, j0, tx, ty),
                   hv = halo_cell<0, 1>(Hvom, i0, j0, tx, ty), hh = halo_cell<1, 1>(Hz, i0, j0, tx, ty);
    // Column cursors (rh_tile.h) that can always be dereferenced: a point outside its array, a thread outside the box or without a halo
    // cell reads an element that exists and whose value nobody uses, so every load of a level is issued unconditionally -- round 2 had
    // an exec-mask branch around each of the 16, and at a join of paths with different numbers of loads in flight hipcc waits for all of
    // them (vmcnt(0)), which pins the prefetch distance to one level.
    const double *ca[RH_NCONS], *cc[RH_NCONS], *qa[RH_NCONS];
#pragma unroll
    for (int n = 0; n < RH_NCONS; n++) {
        A4 an = adv; an.p = adv.p + (long)adv.sn * n;
        ca[n] = col_base_any(an, i, j); cc[n] = col_base_any(cnew, ic, jc, n); qa[n] = halo_base_any(an, ht);
    }
    const double *cHu = col_base_any(Huon, i, j), *cHv = col_base_any(Hvom, i, j), *cHz = col_base_any(Hz, i, j), *cW = col_base_any(W, i, j);
    const double *qHu = halo_base_any(Huon, hu), *qHv = halo_base_any(Hvom, hv), *qHz = halo_base_any(Hz, hh);
    // The operands of level k, requested TWO levels ahead: cnew(k), the halo cell of the advected field, its own point at k + 2 (the new
    // top of the 4-level vertical window), Huon / Hvom / Hz own and halo, W(k + 1).  Two register sets, the loop unrolled by two: a set is
    // consumed at the top of its level (tiles, window, the values the level's arithmetic keeps) and refilled at once for level k + 2;
    // no copy ties a set to the wait of the other, so a wave waits for the older level only.
    struct Lv { double cn[RH_NCONS], th[RH_NCONS], nx[RH_NCONS], Hu, Hv, hz, uh, vh, hh, W1; };
    auto load = [&](int k) -> Lv {
        const int kc = k < N ? k : N, k2 = k + 2 < N ? k + 2 : N, kw = k + 1 < N + 1 ? k + 1 : N + 1;
        Lv v;
#pragma unroll
        for (int n = 0; n < RH_NCONS; n++) { v.cn[n] = cc[n][(unsigned)kc * cnew.sz]; v.th[n] = qa[n][(unsigned)kc * adv.sz]; v.nx[n] = ca[n][(unsigned)k2 * adv.sz]; }
        v.Hu = cHu[(unsigned)kc * Huon.sz]; v.Hv = cHv[(unsigned)kc * Hvom.sz]; v.hz = cHz[(unsigned)kc * Hz.sz];
        v.uh = qHu[(unsigned)kc * Huon.sz]; v.vh = qHv[(unsigned)kc * Hvom.sz]; v.hh = (MODE == 0) ? qHz[(unsigned)kc * Hz.sz] : 0.0;
        v.W1 = cW[(unsigned)kw * W.sz];
        return v;
    };
    double wm1[RH_NCONS], w0[RH_NCONS], wp1[RH_NCONS], wp2[RH_NCONS], FCk[RH_NCONS], cn[RH_NCONS];
#pragma unroll
    for (int n = 0; n < RH_NCONS; n++) { wm1[n] = 0.0; w0[n] = ca[n][0]; wp1[n] = ca[n][(unsigned)(1 < N ? 1 : N) * adv.sz]; wp2[n] = 0.0; FCk[n] = 0.0; cn[n] = 0.0; }
    double Wk = cW[0];
    auto level = [&](int k, Lv &S, int knext) {
        double *set = lds + (k & 1) * SET;
        double *st[RH_NCONS];
#pragma unroll
        for (int n = 0; n < RH_NCONS; n++) { st[n] = set + n * TT::SIZE; tile_put<2, 2>(st[n], tx, ty, w0[n], ht, S.th[n]); }
        double *su = set + RH_NCONS * TT::SIZE, *sv = su + TU::SIZE, *sh = sv + TV::SIZE;
        tile_put<1, 0>(su, tx, ty, S.Hu, hu, S.uh); tile_put<0, 1>(sv, tx, ty, S.Hv, hv, S.vh);
        if (MODE == 0) tile_put<1, 1>(sh, tx, ty, S.hz, hh, S.hh);
        const double Hu0 = S.Hu, Hv0 = S.Hv, hz = S.hz, Wk1 = S.W1;
#pragma unroll
        for (int n = 0; n < RH_NCONS; n++) { cn[n] = S.cn[n]; wp2[n] = S.nx[n]; }
        S = load(knext);                                   // level k + 2 into the set that has just been consumed
        __syncthreads();
        if (act) {
            const double Hu1 = su[(ty)*TU::PX + tx + 2], Hv1 = sv[(ty + 2) * TV::PX + tx];
            const double mu1 = inv[0 * NT], mu0 = inv[1 * NT], mup = inv[2 * NT], mupp = inv[3 * NT];
            const double mv1 = inv[4 * NT], mv0 = inv[5 * NT], mvp = inv[6 * NT], mvpp = inv[7 * NT];
            double DC = 0.0;
            if (MODE == 0) DC = 1.0 / (hz - pmn * (Hu1 - Hu0 + Hv1 - Hv0 + (Wk1 - Wk)));
#pragma unroll
            for (int n = 0; n < RH_NCONS; n++) {
                const double t0 = w0[n];
                const double *c0 = st[n] + (ty + 2) * TT::PX + (tx + 2);
                const double tm2 = c0[-2], tm1 = c0[-1], tp1 = c0[1], tp2 = c0[2];
                const double sm2 = c0[-2 * TT::PX], sm1 = c0[-TT::PX], sp1 = c0[TT::PX], sp2 = c0[2 * TT::PX];
                double FX0, FX1, FE0, FE1;
                if (SCHEME == 2) {
                    FX0 = Hu0 * 0.5 * (tm1 + t0); FX1 = Hu1 * 0.5 * (t0 + tp1);
                    FE0 = Hv0 * 0.5 * (sm1 + t0); FE1 = Hv1 * 0.5 * (t0 + sp1);
                } else if (SCHEME == 0) {
                    FX0 = a3_tflux_u3(tm2, tm1, t0, tp1, mu1, mu0, mup, Hu0, sgn); FX1 = a3_tflux_u3(tm1, t0, tp1, tp2, mu0, mup, mupp, Hu1, sgn);
                    FE0 = a3_tflux_u3(sm2, sm1, t0, sp1, mv1, mv0, mvp, Hv0, sgn); FE1 = a3_tflux_u3(sm1, t0, sp1, sp2, mv0, mvp, mvpp, Hv1, sgn);
                } else {
                    FX0 = a3_tflux_c4(tm2, tm1, t0, tp1, mu1, mu0, mup, Hu0, c4mode); FX1 = a3_tflux_c4(tm1, t0, tp1, tp2, mu0, mup, mupp, Hu1, c4mode);
                    FE0 = a3_tflux_c4(sm2, sm1, t0, sp1, mv1, mv0, mvp, Hv0, c4mode); FE1 = a3_tflux_c4(sm1, t0, sp1, sp2, mv0, mvp, mvpp, Hv1, c4mode);
                }
                const double FCk1 = win_flux_w(wm1[n], w0[n], wp1[n], wp2[n], k + 1, N, Wk1);
                if (MODE == 0) {
                    const double store = hz * (P.cff1 * t0 + P.cff2 * cn[n]) - pmn * (FX1 - FX0 + FE1 - FE0);
                    const double c4 = FCk1 - FCk[n];
                    sst(i, j, k, n) = DC * (store - pmn * c4);
                    double tn = t0 * 0.5 * (hz + hz);
                    if (P.mix_on[n]) {
                        const double *h0 = sh + (ty + 1) * TH::PX + (tx + 1);
                        const double hzxm = h0[-1], hzxp = h0[1], hzym = h0[-TH::PX], hzyp = h0[TH::PX];
                        double GX0 = inv[(8 + 4 * n) * NT] * (hz + hzxm) * (t0 - tm1); GX0 *= mu0;
                        double GX1 = inv[(9 + 4 * n) * NT] * (hzxp + hz) * (tp1 - t0); GX1 *= mup;
                        double GE0 = inv[(10 + 4 * n) * NT] * (hz + hzym) * (t0 - sm1); GE0 *= mv0;
                        double GE1 = inv[(11 + 4 * n) * NT] * (hzyp + hz) * (sp1 - t0); GE1 *= mvp;
                        tn += cmix * ((GX1 - GX0) + (GE1 - GE0));
                    }
                    cnew(i, j, k, n) = tn;
                } else {
                    double x = cn[n];
                    const double cff1 = pmn * (FX1 - FX0);
                    const double cff2 = pmn * (FE1 - FE0);
                    const double cff3 = cff1 + cff2;
                    x -= cff3;
                    const double cff4 = FCk1 - FCk[n];
                    cnew(i, j, k, n) = (x - pmn * cff4) / hz;
                }
                FCk[n] = FCk1;
            }
        }
#pragma unroll
        for (int n = 0; n < RH_NCONS; n++) { wm1[n] = w0[n]; w0[n] = wp1[n]; wp1[n] = wp2[n]; }
        Wk = Wk1;
    };
    // (no test inside the loop: a level that may be skipped would make every wait count after it conservative)
    Lv A = load(0), B = load(1);
    int k = 0;
    for (; k + 1 <= N; k += 2) { level(k, A, k + 2); level(k + 1, B, k + 3); }
    if (k <= N) level(k, A, k + 2);
}

static int tracers_tiled(rh_ctx *c, rh_state *s, int mode, int iic, int ntfirst)
{
    const int N = c->N; const double dt = c->prm.dt;
    rh_box bx = c->bx; bx.lo[2] = 0; bx.hi[2] = N;
    const rh_box g2 = bx_grow(bx, 2, 2, 0), g1 = bx_grow(bx, 1, 1, 0), b2 = bx_slab(bx, 2, 0);
    rh_box bw = bx; bw.hi[2] = N + 1;
    const rh_array4 &adv = mode == 0 ? s->cons_old : s->sstore;
    if (s->cons_old.ncomp < RH_NCONS || s->cons_new.ncomp < RH_NCONS || s->sstore.ncomp < RH_NCONS || s->diff2.ncomp < RH_NCONS)
        return rh_fail("tracers_tiled: tracer arrays need %d components", RH_NCONS);
    RH_NEED(adv, g2); RH_NEED(s->cons_new, bx); RH_NEED(s->sstore, bx); RH_NEED(s->Hz, mode == 0 ? g1 : bx); RH_NEED(c->W, bw);
    RH_NEED(s->Huon, bx_nodal(bx, 0)); RH_NEED(s->Hvom, bx_nodal(bx, 1)); RH_NEED(s->pm, bx_slab(g1, 2, 0)); RH_NEED(s->pn, bx_slab(g1, 2, 0));
    { rh_box m = b2; m.lo[0] -= 1; m.hi[0] += 2; RH_NEED(s->msku, m); }
    { rh_box m = b2; m.lo[1] -= 1; m.hi[1] += 2; RH_NEED(s->mskv, m); }
    if (mode == 0) { rh_box d = bx_slab(g1, 2, 0); d.hi[2] = 1; RH_NEED(s->diff2, d); }
    TrcPar P; memset(&P, 0, sizeof P);
    P.N = N; P.flat = c->prm.flat_bathymetry ? 1 : 0; P.dt = dt;
    if (mode == 0) {
        const double GammaT = 1.0 / 6.0;
        if (iic == ntfirst) { P.cff = 0.5 * dt; P.cff1 = 1.0; P.cff2 = 0.0; }
        else { P.cff = (1.0 - GammaT) * dt; P.cff1 = 0.5 + GammaT; P.cff2 = 0.5 - GammaT; }
        // which (component, level) of diff2 each tracer's t3dmix reads: setup_step.cpp:283-286 + the 3-index access of t3dmix.cpp:34,43
        P.mix_on[0] = 1; P.mix_comp[0] = 0; P.mix_k[0] = 0;
        P.mix_on[1] = c->prm.use_salt ? 1 : 0; P.mix_comp[1] = 0; P.mix_k[1] = 1;
        P.mix_on[2] = 1; P.mix_comp[2] = 2; P.mix_k[2] = 0;
    }
    const int scheme = (mode == 0 && c->prm.flat_bathymetry) ? 2 : (c->prm.tracer_c4 ? 1 : 0);
#define RH_TT(M, S) k_tracers_tiled<M, S><<<rt_grid(bx), rt_block(), 0, c->stream>>>(mkB(bx), mkA(adv), mkA(s->cons_new), mkA(s->sstore), mkA(s->Hz), \
        mkA(s->Huon), mkA(s->Hvom), mkA(c->W), mkA(s->pm), mkA(s->pn), mkA(s->msku), mkA(s->mskv), mkA(s->diff2), P)
    if (mode == 0) { if (scheme == 0) RH_TT(0, 0); else if (scheme == 1) RH_TT(0, 1); else RH_TT(0, 2); }
    else { if (scheme == 0) RH_TT(1, 0); else RH_TT(1, 1); }
#undef RH_TT
    RH_LAUNCH_CHECK(); return 0;
}
int launch_prestep_t3(rh_ctx *c, rh_state *s, int iic, int ntfirst) { TScope t(c, "prestep_t3"); return tracers_tiled(c, s, 0, iic, ntfirst); }
int launch_rhs_t3(rh_ctx *c, rh_state *s) { TScope t(c, "rhs_t3"); return tracers_tiled(c, s, 1, 0, 0); }

// ------------------------------------------------------------------------------------------------------------------
// cons *= mskr, REMORA_advance_3d_ml.cpp:51-62
__global__ void k_mask_cons(B3 bx, A4 cons, A4 mskr, int nc)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const int k = bx.lo2 + RH_KZ;
    const double m = mskr(i, j, 0);
    for (int n = 0; n < nc; n++) cons(i, j, k, n) *= m;
}
int launch_mask_cons(rh_ctx *c, rh_state *s)
{
    RH_NEED(s->cons_new, c->bx); RH_NEED(s->mskr, bx_slab(c->bx, 2, 0));
    k_mask_cons<<<grid3(c->bx), rh_block(), 0, c->stream>>>(mkB(c->bx), mkA(s->cons_new), mkA(s->mskr), RH_NCONS);
    RH_LAUNCH_CHECK(); return 0;
}
