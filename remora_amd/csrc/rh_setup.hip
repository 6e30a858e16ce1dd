// rh_setup.hip -- device kernels of the predictor half of the baroclinic step (REMORA::setup_step and its leaves).
//
// Mapping: one thread per (i,j) column, i along the wavefront (64 consecutive i per wave => 512-B coalesced rows),
// k marched in registers.  Horizontal neighbours are read straight from global memory: consecutive lanes read
// consecutive addresses for every stencil offset, and the +-2 re-reads are served by L1/L2.  Every kernel fuses the
// reference's chain of ParallelFor passes over scratch FABs (FX, FE, curv, grad, FC, DC, UFx, ...) into one sweep
// with the same per-expression operation order (build with -ffp-contract=off), so results are bit-identical to the
// pass-by-pass form.
#include "rh_common.h"
#include "rh_tile.h"

// ------------------------------------------------------------------------------------------------------------------
// bottom stress, REMORA_setup_step.cpp:115-137
__global__ void k_bstress(B3 ub, B3 vb, B3 un, A4 rdrag, A4 bustr, A4 bvstr, A4 uold, A4 vold)
{
    RH_IJ(un);
    if (inb2(ub, i, j)) bustr(i, j, 0) = 0.5 * (rdrag(i - 1, j, 0) + rdrag(i, j, 0)) * (uold(i, j, 0));
    if (inb2(vb, i, j)) bvstr(i, j, 0) = 0.5 * (rdrag(i, j - 1, 0) + rdrag(i, j, 0)) * (vold(i, j, 0));
}
int launch_bstress(rh_ctx *c, rh_state *s)
{
    const rh_box b2 = bx_slab(c->bx, 2, 0);
    const rh_box ub = bx_grow(bx_nodal(b2, 0), 1, 1, 0), vb = bx_grow(bx_nodal(b2, 1), 1, 1, 0), un = bx_union(ub, vb);
    RH_NEED(s->rdrag, bx_grow(b2, 1, 1, 0)); RH_NEED(s->bustr, ub); RH_NEED(s->bvstr, vb);
    RH_NEED(s->xvel_old, ub); RH_NEED(s->yvel_old, vb);
    k_bstress<<<grid2(un), rh_block(), 0, c->stream>>>(mkB(ub), mkB(vb), mkB(un), mkA(s->rdrag), mkA(s->bustr), mkA(s->bvstr),
                                                       mkA(s->xvel_old), mkA(s->yvel_old));
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Huon / Hvom from the old velocities, REMORA_setup_step.cpp:187-197
__global__ void k_massflux_old(B3 bu, B3 bv, B3 un, A4 Hz, A4 Huon, A4 Hvom, A4 uold, A4 vold, A4 pm, A4 pn)
{
    RH_IJ(un);
    const int k = un.lo2 + RH_KZ;
    if (inb2(bu, i, j)) { const double on_u = 2.0 / (pn(i - 1, j, 0) + pn(i, j, 0)); Huon(i, j, k) = 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k)) * uold(i, j, k) * on_u; }
    if (inb2(bv, i, j)) { const double om_v = 2.0 / (pm(i, j - 1, 0) + pm(i, j, 0)); Hvom(i, j, k) = 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k)) * vold(i, j, k) * om_v; }
}
int launch_massflux_old(rh_ctx *c, rh_state *s)
{
    const rh_box bu = bx_of(s->Huon), bv = bx_of(s->Hvom), un = bx_union(bu, bv);
    RH_NEED(s->Hz, bx_growlo_i(bu)); RH_NEED(s->Hz, bx_growlo_j(bv)); RH_NEED(s->xvel_old, bu); RH_NEED(s->yvel_old, bv);
    RH_NEED(s->pn, bx_slab(bx_growlo_i(bu), 2, 0)); RH_NEED(s->pm, bx_slab(bx_growlo_j(bv), 2, 0));
    k_massflux_old<<<grid3(un), rh_block(), 0, c->stream>>>(mkB(bu), mkB(bv), mkB(un), mkA(s->Hz), mkA(s->Huon), mkA(s->Hvom),
                                                            mkA(s->xvel_old), mkA(s->yvel_old), mkA(s->pm), mkA(s->pn));
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::rho_eos, REMORA_rho_eos.cpp:22-99: linear EOS, vertical averages rhoA / rhoS, Brunt-Vaisala frequency.
// One downward sweep per column (the reference makes three passes and re-reads rho twice).
__global__ void k_rho_eos(B3 bx, A4 st, A4 rho, A4 rhoA, A4 rhoS, A4 bvf, A4 Hz, A4 z_w, A4 z_r, A4 h, A4 mskr, int N,
                          double R0, double S0, double T0, double Tcoef, double Scoef, double rho0, double g)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const double m = mskr(i, j, 0);
    const double gorho0 = g / rho0;
    double r_up = 0.0, z_up = 0.0, A = 0.0, S = 0.0;
    for (int k = N; k >= 0; k--) {
        const double r = (R0 - R0 * Tcoef * (st(i, j, k, 0) - T0) + R0 * Scoef * (st(i, j, k, 1) - S0) - 1000.0) * m;
        rho(i, j, k) = r;
        const double hz = Hz(i, j, k), zr = z_r(i, j, k);
        if (k == N) { const double cff0 = r * hz; S = 0.5 * cff0 * hz; A = cff0; }
        else {
            const double cff1 = r * hz; S += hz * (A + 0.5 * cff1); A += cff1;
            bvf(i, j, k + 1) = -gorho0 * (r_up - r) / (z_up - zr);
        }
        r_up = r; z_up = zr;
    }
    const double cff2 = 1.0 / rho0;
    const double cff11 = 1.0 / (z_w(i, j, N + 1) + h(i, j, 0, 0));
    A *= cff2 * cff11;
    S *= 2.0 * cff11 * cff11 * cff2;
    rhoA(i, j, 0) = A; rhoS(i, j, 0) = S;
}
extern "C" int rh_rho_eos(rh_ctx *c, rh_box bx, rh_array4 state, rh_array4 rho, rh_array4 rhoA, rh_array4 rhoS, rh_array4 bvf,
                          rh_array4 Hz, rh_array4 z_w, rh_array4 z_r, rh_array4 h, rh_array4 mskr, int N)
{
    TScope t(c, "rho_eos");
    const rh_box b2 = bx_slab(bx, 2, 0);
    rh_box bw = bx; bw.lo[2] = 0; bw.hi[2] = N + 1;
    bx.lo[2] = 0; bx.hi[2] = N;
    if (state.ncomp < 2) return rh_fail("rh_rho_eos: state needs temp and salt");
    RH_NEED(state, bx); RH_NEED(rho, bx); RH_NEED(Hz, bx); RH_NEED(z_r, bx); RH_NEED(z_w, bw); RH_NEED(bvf, bw);
    RH_NEED(rhoA, b2); RH_NEED(rhoS, b2); RH_NEED(h, b2); RH_NEED(mskr, b2);
    const rh_params &P = c->prm;
    k_rho_eos<<<grid2(bx), rh_block(), 0, c->stream>>>(mkB(bx), mkA(state), mkA(rho), mkA(rhoA), mkA(rhoS), mkA(bvf), mkA(Hz),
        mkA(z_w), mkA(z_r), mkA(h), mkA(mskr), N, P.R0, P.S0, P.T0, P.Tcoef, P.Scoef, P.rho0, P.g);
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// analytic vertical mixing = init_custom_vmix of Exec/<case>/prob.cpp (Upwelling/prob.cpp:163-185, Seamount/prob.cpp:167-190,
// DoubleGyre/prob.cpp:116-138), which REMORA::set_analytical_vmix (REMORA.cpp:565-573) calls every step
__global__ void k_vmix(B3 b, A4 Akv, A4 Akt, A4 z_w, int mode, double akv0, double akt0, double akt1, double akt2)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const int k = b.lo2 + RH_KZ;
    if (mode == RH_VMIX_UPWELLING) {
        Akv(i, j, k) = 2.0e-03 + 8.0e-03 * exp(z_w(i, j, k) / 150.0);
        Akt(i, j, k, 0) = 1.0e-6; Akt(i, j, k, 1) = 1.0e-6; Akt(i, j, k, 2) = 0.0;
    } else {
        Akv(i, j, k) = akv0; Akt(i, j, k, 0) = akt0; Akt(i, j, k, 1) = akt1; Akt(i, j, k, 2) = akt2;
    }
}
int launch_vmix(rh_ctx *c, rh_state *s)
{
    rh_box b = bx_of(s->Akv);
    if (s->Akt.ncomp < RH_NCONS) return rh_fail("vmix: Akt needs %d components", RH_NCONS);
    RH_NEED(s->Akt, b);
    if (c->prm.vmix_mode == RH_VMIX_UPWELLING) RH_NEED(s->z_w, b);
    k_vmix<<<grid3(b), rh_block(), 0, c->stream>>>(mkB(b), mkA(s->Akv), mkA(s->Akt), mkA(s->z_w), c->prm.vmix_mode,
        c->prm.Akv_const, c->prm.Akt_const[0], c->prm.Akt_const[1], c->prm.Akt_const[2]);
    RH_LAUNCH_CHECK(); return 0;
}

// set_zeta_to_Ztavg (REMORA_setup_step.cpp:208): zeta comps 0..2 := Zt_avg1
__global__ void k_set_zeta(B3 b, A4 zeta, A4 Zt)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const double z = Zt(i, j, 0);
    zeta(i, j, 0, 0) = z; zeta(i, j, 0, 1) = z; zeta(i, j, 0, 2) = z;
}
int launch_set_zeta(rh_ctx *c, rh_state *s)
{
    const rh_box b = bx_grow(bx_slab(c->bx, 2, 0), 3, 3, 0);
    RH_NEED(s->zeta, b); RH_NEED(s->Zt_avg1, b);
    if (s->zeta.ncomp < 3) return rh_fail("set_zeta: zeta needs 3 components");
    k_set_zeta<<<grid2(b), rh_block(), 0, c->stream>>>(mkB(b), mkA(s->zeta), mkA(s->Zt_avg1));
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// omega: REMORA_prestep_t_advection.cpp:97-125 == REMORA_advance_3d.cpp:272-293.  Upward integration of the horizontal
// flux divergence, then removal of the part proportional to (z_w + h).
__global__ void k_omega(B3 b, A4 W, A4 Huon, A4 Hvom, A4 z_w, A4 h, int N)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    double w = 0.0;
    W(i, j, 0) = 0.0;
    for (int k = 1; k <= N + 1; k++) {
        w = w - (Huon(i + 1, j, k - 1) - Huon(i, j, k - 1)) - (Hvom(i, j + 1, k - 1) - Hvom(i, j, k - 1));
        W(i, j, k) = w;
    }
    const double hh = h(i, j, 0, 0);
    const double wrk = w / (z_w(i, j, N + 1) + hh);
    for (int k = 1; k <= N; k++) W(i, j, k) = W(i, j, k) - wrk * (z_w(i, j, k) + hh);
    W(i, j, N + 1) = 0.0;
}
// Same sweep with the running integral w(k) parked in LDS ([k][lane], one wavefront per block, one column per lane) instead of
// being written to W and read back, and with the operands of level k+D requested while level k is integrated (register ring,
// see k_column_solver in rh_adv3d.hip).  Huon(i+1) comes from the neighbouring lane.
#define OM_D 12
template <int D>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_omega_lds(B3 b, A4 W, A4 Huon, A4 Hvom, A4 z_w, A4 h, int N)
{
    extern __shared__ double om_lds[];
    const int lane = threadIdx.x;
    double *sW = om_lds + lane;                                       // sW[k * 64], k = 1 .. N+1
    // waves overlap by one column: lane 63 only carries Huon(i+1) of lane 62 into the shuffle (no one-element loads); tiles are
    // walked in 1 x 4 patches so that Hvom(j+1), the own row of the wave above, is found in the XCD's L2
    const RhTile rh_t_ = rh_tile_patch<1, 4>();
    const int i = b.lo0 + rh_t_.x * 63 + lane, j = b.lo1 + rh_t_.y;
    const bool act = lane < 63 && i <= b.hi0;
    const int ic = i <= b.hi0 ? i : b.hi0;
    const int iu = i <= b.hi0 + 1 ? i : b.hi0 + 1;                    // Huon lives on the faces b.lo0 .. b.hi0 + 1
    struct L { double u, v0, v1; };
    auto load = [&](int kk) -> L {                                    // level kk (clamped) of the fluxes
        const int kc = kk < N ? kk : N;
        L r; r.u = Huon(iu, j, kc); r.v0 = Hvom(ic, j, kc); r.v1 = Hvom(ic, j + 1, kc); return r;
    };
    double w = 0.0;
    auto step = [&](int k, const L &c) {                              // :102-109: W(k) = W(k-1) - dHuon(k-1) - dHvom(k-1)
        const double up = __shfl_down(c.u, 1);
        w = w - (up - c.u) - (c.v1 - c.v0);
        sW[k * 64] = w;
    };
    L ring[D];
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = load(u);
    int kb = 1;
    for (; kb + D - 1 <= N + 1; kb += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            step(kb + u, ring[u]);
            __builtin_amdgcn_sched_barrier(0);
            ring[u] = load(kb + u - 1 + D);
        }
    }
#pragma unroll
    for (int u = 0; u < D; u++) if (kb + u <= N + 1) step(kb + u, ring[u]);
    // ---- remove the part proportional to (z_w + h), :111-125
    const double hh = h(ic, j, 0, 0);
    const double wrk = w / (z_w(ic, j, N + 1) + hh);
    if (act) { W(i, j, 0) = 0.0; W(i, j, N + 1) = 0.0; }
    double zr[D];
#pragma unroll
    for (int u = 0; u < D; u++) { const int k = 1 + u; zr[u] = z_w(ic, j, k < N ? k : N); }
    kb = 1;
    for (; kb + D - 1 <= N; kb += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            const int k = kb + u;
            const double x = sW[k * 64] - wrk * (zr[u] + hh);
            if (act) W(i, j, k) = x;
            __builtin_amdgcn_sched_barrier(0);
            const int kn = k + D; zr[u] = z_w(ic, j, kn < N ? kn : N);
        }
    }
#pragma unroll
    for (int u = 0; u < D; u++) { const int k = kb + u; if (k <= N) { const double x = sW[k * 64] - wrk * (zr[u] + hh); if (act) W(i, j, k) = x; } }
}
// setup_step's pair "mass fluxes of the old velocities" (:187-197) + omega (prestep_t_advection.cpp:97-125) in one sweep: the wave that
// integrates a column forms Huon / Hvom of its faces from uold, vold, Hz (k_massflux_old's expressions), stores them, and feeds them
// to the integration from registers, so the two flux arrays are not read back (k_massflux_old + k_omega_lds moved 11.4 doubles per
// cell, this moves ~7).  It covers the faces omega touches (u faces b.lo0 .. b.hi0+1, v faces b.lo1 .. b.hi1+1 of the rows / columns
// of b); the ghost faces outside are left to k_massflux_old on four strips per array (launch_massflux_omega).
#define MO_D 6
template <int D>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_massflux_omega(B3 b, A4 W, A4 Huon, A4 Hvom, A4 uold, A4 vold, A4 Hz, A4 pm, A4 pn, A4 z_w, A4 h, int N)
{
    extern __shared__ double om_lds[];
    const int lane = threadIdx.x;
    double *sW = om_lds + lane;                                       // sW[k * 64], k = 1 .. N+1
    const RhTile rh_t_ = rh_tile_patch<1, 4>();
    const int i = b.lo0 + rh_t_.x * 63 + lane, j = b.lo1 + rh_t_.y;
    const bool act = lane < 63 && i <= b.hi0;
    const int ic = i <= b.hi0 ? i : b.hi0;
    const int iu = i <= b.hi0 + 1 ? i : b.hi0 + 1;
    const bool st_u = i <= b.hi0 + 1 && (lane < 63 || i == b.hi0 + 1);     // this lane owns the u face iu (lane 63's is the next wave's lane 0)
    const bool st_v1 = act && j == b.hi1;                             // the top row also owns its upper v faces
    const double on_u = 2.0 / (pn(iu - 1, j, 0) + pn(iu, j, 0));
    const double om_v0 = 2.0 / (pm(ic, j - 1, 0) + pm(ic, j, 0)), om_v1 = 2.0 / (pm(ic, j, 0) + pm(ic, j + 1, 0));
    struct L { double u, hu0, hu1, v0, v1, hm, hp; };
    auto load = [&](int kk) -> L {                                    // level kk (clamped) of the operands
        const int kc = kk < N ? kk : N;
        L r; r.u = uold(iu, j, kc); r.hu0 = Hz(iu, j, kc); r.hu1 = Hz(iu - 1, j, kc);
        r.v0 = vold(ic, j, kc); r.v1 = vold(ic, j + 1, kc); r.hm = Hz(ic, j - 1, kc); r.hp = Hz(ic, j + 1, kc);
        return r;
    };
    double w = 0.0;
    auto step = [&](int k, const L &c) {                              // fluxes of level k-1, then W(k) = W(k-1) - dHuon(k-1) - dHvom(k-1)
        const double hu = 0.5 * (c.hu0 + c.hu1) * c.u * on_u;
        const double hv0 = 0.5 * (c.hu0 + c.hm) * c.v0 * om_v0;       // for the lanes that use it (i <= b.hi0) iu == ic: hu0 is Hz(ic, j)
        const double hv1 = 0.5 * (c.hp + c.hu0) * c.v1 * om_v1;
        if (st_u) Huon(iu, j, k - 1) = hu;
        if (act) Hvom(ic, j, k - 1) = hv0;
        if (st_v1) Hvom(ic, j + 1, k - 1) = hv1;
        const double up = __shfl_down(hu, 1);
        w = w - (up - hu) - (hv1 - hv0);
        sW[k * 64] = w;
    };
    L ring[D];
#pragma unroll
    for (int u = 0; u < D; u++) ring[u] = load(u);
    int kb = 1;
    for (; kb + D - 1 <= N + 1; kb += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            step(kb + u, ring[u]);
            __builtin_amdgcn_sched_barrier(0);
            ring[u] = load(kb + u - 1 + D);
        }
    }
#pragma unroll
    for (int u = 0; u < D; u++) if (kb + u <= N + 1) step(kb + u, ring[u]);
    // ---- remove the part proportional to (z_w + h), :111-125
    const double hh = h(ic, j, 0, 0);
    const double wrk = w / (z_w(ic, j, N + 1) + hh);
    if (act) { W(i, j, 0) = 0.0; W(i, j, N + 1) = 0.0; }
    double zr[D];
#pragma unroll
    for (int u = 0; u < D; u++) { const int k = 1 + u; zr[u] = z_w(ic, j, k < N ? k : N); }
    kb = 1;
    for (; kb + D - 1 <= N; kb += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            const int k = kb + u;
            const double x = sW[k * 64] - wrk * (zr[u] + hh);
            if (act) W(i, j, k) = x;
            __builtin_amdgcn_sched_barrier(0);
            const int kn = k + D; zr[u] = z_w(ic, j, kn < N ? kn : N);
        }
    }
#pragma unroll
    for (int u = 0; u < D; u++) { const int k = kb + u; if (k <= N) { const double x = sW[k * 64] - wrk * (zr[u] + hh); if (act) W(i, j, k) = x; } }
}
// *applied = 0: the caller runs launch_massflux_old and rh_omega
int launch_massflux_omega(rh_ctx *c, rh_state *s, rh_box gbx1, int N, int *applied)
{
    *applied = 0;
    rh_box b = gbx1; b.lo[2] = 0; b.hi[2] = N; rh_box bw = b; bw.hi[2] = N + 1;
    const size_t lds_bytes = sizeof(double) * (size_t)(N + 2) * 64;
    if (c->unfused || lds_bytes > RH_LDS_MAX) return 0;
    const rh_box bu = bx_of(s->Huon), bv = bx_of(s->Hvom);
    rh_box cu = bx_nodal(b, 0), cv = bx_nodal(b, 1);                  // the faces the fused kernel writes
    if (bx_and(bu, cu).lo[0] != cu.lo[0] || bx_and(bu, cu).hi[0] != cu.hi[0] || bx_and(bu, cu).lo[1] != cu.lo[1] || bx_and(bu, cu).hi[1] != cu.hi[1] ||
        bx_and(bv, cv).lo[0] != cv.lo[0] || bx_and(bv, cv).hi[0] != cv.hi[0] || bx_and(bv, cv).lo[1] != cv.lo[1] || bx_and(bv, cv).hi[1] != cv.hi[1] ||
        bu.lo[2] != 0 || bu.hi[2] != N || bv.lo[2] != 0 || bv.hi[2] != N) return 0;
    TScope t(c, "massflux_omega");
    RH_NEED(c->W, bw); RH_NEED(s->z_w, bw); RH_NEED(s->h, bx_slab(b, 2, 0));
    RH_NEED(s->Hz, bx_growlo_i(bu)); RH_NEED(s->Hz, bx_growlo_j(bv)); RH_NEED(s->xvel_old, bu); RH_NEED(s->yvel_old, bv);
    RH_NEED(s->pn, bx_slab(bx_growlo_i(bu), 2, 0)); RH_NEED(s->pm, bx_slab(bx_growlo_j(bv), 2, 0));
    RH_NEED(s->Hz, bx_grow(b, 1, 1, 0)); RH_NEED(s->pm, bx_slab(bx_grow(b, 1, 1, 0), 2, 0)); RH_NEED(s->pn, bx_slab(bx_grow(b, 1, 1, 0), 2, 0));
    // ghost faces outside the fused kernel's reach: four strips per flux array
    const rh_box none = bx_make(0, 0, 0, -1, -1, -1);
    auto strips = [&](const rh_box &arr, const rh_box &cov, bool is_u) -> int {
        rh_box q[4] = {arr, arr, arr, arr};
        q[0].hi[1] = cov.lo[1] - 1;                                                   // rows below
        q[1].lo[1] = cov.hi[1] + 1;                                                   // rows above
        q[2].lo[1] = cov.lo[1]; q[2].hi[1] = cov.hi[1]; q[2].hi[0] = cov.lo[0] - 1;   // columns left
        q[3].lo[1] = cov.lo[1]; q[3].hi[1] = cov.hi[1]; q[3].lo[0] = cov.hi[0] + 1;   // columns right
        for (const rh_box &r : q) {
            if (r.lo[0] > r.hi[0] || r.lo[1] > r.hi[1]) continue;
            k_massflux_old<<<grid3(r), rh_block(), 0, c->stream>>>(mkB(is_u ? r : none), mkB(is_u ? none : r), mkB(r), mkA(s->Hz), mkA(s->Huon), mkA(s->Hvom),
                                                                   mkA(s->xvel_old), mkA(s->yvel_old), mkA(s->pm), mkA(s->pn));
            RH_LAUNCH_CHECK();
        }
        return 0;
    };
    RH_TRY(strips(bu, cu, true)); RH_TRY(strips(bv, cv, false));
    RH_HIP(rh_lds_optin(k_massflux_omega<MO_D>, lds_bytes));
    const dim3 grid((b.hi[0] - b.lo[0] + 63) / 63, b.hi[1] - b.lo[1] + 1, 1);         // waves overlap by one column
    k_massflux_omega<MO_D><<<grid, 64, lds_bytes, c->stream>>>(mkB(b), mkA(c->W), mkA(s->Huon), mkA(s->Hvom), mkA(s->xvel_old), mkA(s->yvel_old), mkA(s->Hz),
                                                              mkA(s->pm), mkA(s->pn), mkA(s->z_w), mkA(s->h), N);
    RH_LAUNCH_CHECK();
    *applied = 1; return 0;
}
extern "C" int rh_omega(rh_ctx *c, rh_box gbx1, rh_array4 W, rh_array4 Huon, rh_array4 Hvom, rh_array4 z_w, rh_array4 h, int N)
{
    TScope t(c, "omega");
    rh_box b = gbx1; b.lo[2] = 0; b.hi[2] = N; rh_box bw = b; bw.hi[2] = N + 1;
    RH_NEED(W, bw); RH_NEED(z_w, bw); RH_NEED(Huon, bx_nodal(b, 0)); RH_NEED(Hvom, bx_nodal(b, 1)); RH_NEED(h, bx_slab(b, 2, 0));
    const size_t lds_bytes = sizeof(double) * (size_t)(N + 2) * 64;
    if (!c->unfused && lds_bytes <= RH_LDS_MAX) {
        RH_HIP(rh_lds_optin(k_omega_lds<OM_D>, lds_bytes));
        const dim3 grid((b.hi[0] - b.lo[0] + 63) / 63, b.hi[1] - b.lo[1] + 1, 1);     // waves overlap by one column
        k_omega_lds<OM_D><<<grid, 64, lds_bytes, c->stream>>>(mkB(b), mkA(W), mkA(Huon), mkA(Hvom), mkA(z_w), mkA(h), N);
    } else
        k_omega<<<grid2(b), rh_block(), 0, c->stream>>>(mkB(b), mkA(W), mkA(Huon), mkA(Hvom), mkA(z_w), mkA(h), N);
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// horizontal tracer flux through the x-face i (dir 0) or the y-face j (dir 1) of one level: third-order upstream or
// fourth-order centred (REMORA_prestep_t_advection.cpp:161-266, REMORA_rhs_t_3d.cpp:87-276).
//   t2,t1 = values at face-2, face-1 (below the face), t0,tp = face, face+1 (above); m1,m0,mp = face masks at face-1, face, face+1
//   sgn = -1 (prestep, and rhs_t_3d with bathymetry) or +1 (rhs_t_3d, flat bathymetry: REMORA_rhs_t_3d.cpp:124-125,138-139)
__device__ __forceinline__ double tflux_u3(double t2, double t1, double t0, double tp, double m1, double m0, double mp, double H, double sgn)
{
    const double g1 = (t1 - t2) * m1, g0 = (t0 - t1) * m0, gp = (tp - t0) * mp;
    const double curv0 = -g0 + gp, curv1 = -g1 + g0;              // curv at cell face (=i) and at face-1
    const double maxH = dmax_(H, 0.0), minH = dmin_(H, 0.0);
    const double a = H * 0.5 * (t0 + t1), b = (1.0 / 6.0) * (curv0 * minH + curv1 * maxH);
    return sgn < 0.0 ? a - b : a + b;
}
// mode 0: H*0.5*(t0+t1-cffb*(grad0-grad1))   (prestep and rhs_t_3d with bathymetry)
// mode 1: H*0.5*(t0+t1)+cffb*(grad0+grad1)   (rhs_t_3d, flat bathymetry: REMORA_rhs_t_3d.cpp:138-139)
__device__ __forceinline__ double tflux_c4(double t2, double t1, double t0, double tp, double m1, double m0, double mp, double H, int mode)
{
    const double g1 = (t1 - t2) * m1, g0 = (t0 - t1) * m0, gp = (tp - t0) * mp;
    const double grad0 = 0.5 * (g0 + gp), grad1 = 0.5 * (g1 + g0);
    const double cffb = 1.0 / 3.0;
    return mode == 0 ? H * 0.5 * (t0 + t1 - cffb * (grad0 - grad1)) : H * 0.5 * (t0 + t1) + cffb * (grad0 + grad1);
}
// fourth-order vertical flux through k-face kk (REMORA_prestep_t_advection.cpp:304-332, REMORA_rhs_t_3d.cpp:301-326)
__device__ __forceinline__ double tflux_w(const A4 &t, int i, int j, int kk, int N, double w)
{
    const double c1 = 0.5, c2 = 7.0 / 12.0, c3 = 1.0 / 12.0;
    if (kk >= 2 && kk <= N - 1) return (c2 * (t(i, j, kk - 1) + t(i, j, kk)) - c3 * (t(i, j, kk - 2) + t(i, j, kk + 1))) * w;
    if (kk == N + 1) return 0.0;
    if (kk == N) return (c2 * t(i, j, N - 1) + c1 * t(i, j, N) - c3 * t(i, j, N - 2)) * w;
    if (kk == 1) return (c2 * t(i, j, 1) + c1 * t(i, j, 0) - c3 * t(i, j, 2)) * w;
    return 0.0;
}

// REMORA::prestep_t_advection after omega (REMORA_prestep_t_advection.cpp:127-361) fused with the tracer call of
// REMORA::prestep_diffusion (REMORA_prestep_diffusion.cpp:19-165 with lambda = 1, zero surface/bottom tracer flux:
// REMORA_prestep.cpp:101,166-170), one tracer.  The reference's 5-component cache copy (REMORA_prestep.cpp:63-70) is not needed:
// tnew is read (as tempcache) and overwritten at the same point only.
template <int SCHEME>   // 0 = upstream3, 1 = centered4, 2 = flat-bathymetry centred flux (REMORA_prestep_t_advection.cpp:147-157)
__global__ void k_prestep_t(B3 bx, A4 told, A4 tnew, A4 ts, A4 Hz, A4 Huon, A4 Hvom, A4 W, A4 pm, A4 pn, A4 msku, A4 mskv,
                            int N, double cff, double cff1, double cff2)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const double pmn = cff * pm(i, j, 0) * pn(i, j, 0);
    const double mu1 = msku(i - 1, j, 0), mu0 = msku(i, j, 0), mup = msku(i + 1, j, 0), mupp = msku(i + 2, j, 0);
    const double mv1 = mskv(i, j - 1, 0), mv0 = mskv(i, j, 0), mvp = mskv(i, j + 1, 0), mvpp = mskv(i, j + 2, 0);
    double FCk = 0.0, Wk = W(i, j, 0);
    for (int k = 0; k <= N; k++) {
        const double t0 = told(i, j, k);
        const double tm2 = told(i - 2, j, k), tm1 = told(i - 1, j, k), tp1 = told(i + 1, j, k), tp2 = told(i + 2, j, k);
        const double sm2 = told(i, j - 2, k), sm1 = told(i, j - 1, k), sp1 = told(i, j + 1, k), sp2 = told(i, j + 2, k);
        const double Hu0 = Huon(i, j, k), Hu1 = Huon(i + 1, j, k), Hv0 = Hvom(i, j, k), Hv1 = Hvom(i, j + 1, k);
        double FX0, FX1, FE0, FE1;
        if (SCHEME == 2) {
            FX0 = Hu0 * 0.5 * (tm1 + t0); FX1 = Hu1 * 0.5 * (t0 + tp1);
            FE0 = Hv0 * 0.5 * (sm1 + t0); FE1 = Hv1 * 0.5 * (t0 + sp1);
        } else if (SCHEME == 0) {
            FX0 = tflux_u3(tm2, tm1, t0, tp1, mu1, mu0, mup, Hu0, -1.0); FX1 = tflux_u3(tm1, t0, tp1, tp2, mu0, mup, mupp, Hu1, -1.0);
            FE0 = tflux_u3(sm2, sm1, t0, sp1, mv1, mv0, mvp, Hv0, -1.0); FE1 = tflux_u3(sm1, t0, sp1, sp2, mv0, mvp, mvpp, Hv1, -1.0);
        } else {
            FX0 = tflux_c4(tm2, tm1, t0, tp1, mu1, mu0, mup, Hu0, 0); FX1 = tflux_c4(tm1, t0, tp1, tp2, mu0, mup, mupp, Hu1, 0);
            FE0 = tflux_c4(sm2, sm1, t0, sp1, mv1, mv0, mvp, Hv0, 0); FE1 = tflux_c4(sm1, t0, sp1, sp2, mv0, mvp, mvpp, Hv1, 0);
        }
        const double hz = Hz(i, j, k);
        double store = hz * (cff1 * t0 + cff2 * tnew(i, j, k)) - pmn * (FX1 - FX0 + FE1 - FE0);
        const double Wk1 = W(i, j, k + 1);
        const double FCk1 = tflux_w(told, i, j, k + 1, N, Wk1);
        const double DC = 1.0 / (hz - pmn * (Hu1 - Hu0 + Hv1 - Hv0 + (Wk1 - Wk)));
        const double c4 = FCk1 - FCk;
        ts(i, j, k) = DC * (store - pmn * c4);
        tnew(i, j, k) = t0 * 0.5 * (hz + hz);
        FCk = FCk1; Wk = Wk1;
    }
}
extern "C" int rh_prestep_t(rh_ctx *c, rh_box tbx, rh_array4 told, rh_array4 tnew, rh_array4 tempstore, rh_array4 Hz,
                            rh_array4 Huon, rh_array4 Hvom, rh_array4 W, rh_array4 pm, rh_array4 pn, rh_array4 msku, rh_array4 mskv,
                            int iic, int ntfirst, int N, double dt)
{
    TScope t(c, "prestep_t");
    tbx.lo[2] = 0; tbx.hi[2] = N;
    const rh_box g2 = bx_grow(tbx, 2, 2, 0), b2 = bx_slab(tbx, 2, 0);
    rh_box bw = tbx; bw.hi[2] = N + 1;
    RH_NEED(told, g2); RH_NEED(tnew, tbx); RH_NEED(tempstore, tbx); RH_NEED(Hz, tbx); RH_NEED(W, bw);
    RH_NEED(Huon, bx_nodal(tbx, 0)); RH_NEED(Hvom, bx_nodal(tbx, 1)); RH_NEED(pm, b2); RH_NEED(pn, b2);
    { rh_box m = b2; m.lo[0] -= 1; m.hi[0] += 2; RH_NEED(msku, m); }
    { rh_box m = b2; m.lo[1] -= 1; m.hi[1] += 2; RH_NEED(mskv, m); }
    const double GammaT = 1.0 / 6.0;
    double cff, cff1, cff2;
    if (iic == ntfirst) { cff = 0.5 * dt; cff1 = 1.0; cff2 = 0.0; }
    else { cff = (1.0 - GammaT) * dt; cff1 = 0.5 + GammaT; cff2 = 0.5 - GammaT; }
    const int scheme = c->prm.flat_bathymetry ? 2 : (c->prm.tracer_c4 ? 1 : 0);
#define RH_PT(S) k_prestep_t<S><<<grid2(tbx), rh_block(), 0, c->stream>>>(mkB(tbx), mkA(told), mkA(tnew), mkA(tempstore), mkA(Hz), \
        mkA(Huon), mkA(Hvom), mkA(W), mkA(pm), mkA(pn), mkA(msku), mkA(mskv), N, cff, cff1, cff2)
    if (scheme == 0) RH_PT(0); else if (scheme == 1) RH_PT(1); else RH_PT(2);
#undef RH_PT
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::prestep_diffusion for u (ioff = 1) or v (joff = 1), REMORA_prestep_diffusion.cpp:19-165 with lambda = 1
// (REMORA_prestep.cpp:101): vel*Hz + surface/bottom stress + Adams-Bashforth weighting of the stored right-hand sides,
// including the in-place component swap of rvel (:128-131,154-161).
__global__ void k_prestep_uv(B3 bx, B3 dead, int ioff, int joff, A4 vel, A4 vel_old, A4 rvel, A4 Hz, A4 sstr, A4 bstr, A4 pm, A4 pn,
                             int step, int nrhs, int N, double dt)
{
    // `dead`: points whose rvel(nrhs) is overwritten before anything reads it (setup_step: prsgrd assigns ru / rv(nrhs) on the valid
    // faces right after prestep, REMORA_prsgrd.cpp:134-149,181-196); the store is skipped there.  Empty for the stand-alone leaf.
    // no level depends on another (lambda = 1 removes the vertical flux divergence): one thread per point, grid.z walks k
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const int k = bx.lo2 + RH_KZ;
    const int ii = i - ioff, jj = j - joff;
    const int indx = nrhs ? 0 : 1;
    const double a = vel_old(i, j, k) * 0.5 * (Hz(i, j, k) + Hz(ii, jj, k));
    // FC(k) = dt*(1-lambda)*... = 0 for k < N, FC(N) = dt*sstr
    double b;
    if (k == 0) b = 0.0 - dt * bstr(i, j, 0); else if (k == N) b = dt * sstr(i, j, 0) - 0.0; else b = 0.0 - 0.0;
    if (step == 0) { vel(i, j, k) = a + b; return; }
    const double DC = 0.25 * dt * (pm(i, j, 0) + pm(ii, jj, 0)) * (pn(i, j, 0) + pn(ii, jj, 0));
    const double r_indx = rvel(i, j, k, indx), r_nrhs = rvel(i, j, k, nrhs);
    rvel(i, j, k, indx) = r_nrhs;                                     // swap
    const bool keep = !inb2(dead, i, j);
    if (step == 1) {
        if (keep) rvel(i, j, k, nrhs) = r_indx;
        const double cff3 = 0.5 * DC;
        vel(i, j, k) = a - cff3 * r_nrhs + b;
    } else {
        const double cff1 = 5.0 / 12.0, cff2 = 16.0 / 12.0;
        vel(i, j, k) = a + DC * (cff1 * r_indx - cff2 * r_nrhs) + b;
        if (keep) rvel(i, j, k, nrhs) = 0.0;
    }
}
// u and v in one launch (the driver's form): both run over the same box, so Hz is fetched once for the two (1 of the 12 doubles the
// two launches moved from HBM: 4.28 -> 4.06 ms).  PUV_KC levels per thread: keeping a column for 6 levels also saves the per-level
// re-reads of pm / pn, but those hit in L2 / Infinity Cache anyway and the longer threads ran slower (4.28 ms), so it stays 1.
// Same arithmetic as k_prestep_uv.
#define PUV_KC 1
__device__ __forceinline__ void prestep_uv_point(const A4 &vel, const A4 &vel_old, const A4 &rvel, int i, int j, int k, double hz0, double hz1, double DC,
                                                 double sst, double bst, bool keep, int step, int nrhs, int indx, int N, double dt)
{
    const double a = vel_old(i, j, k) * 0.5 * (hz0 + hz1);
    double b;
    if (k == 0) b = 0.0 - dt * bst; else if (k == N) b = dt * sst - 0.0; else b = 0.0 - 0.0;
    if (step == 0) { vel(i, j, k) = a + b; return; }
    const double r_indx = rvel(i, j, k, indx), r_nrhs = rvel(i, j, k, nrhs);
    rvel(i, j, k, indx) = r_nrhs;                                     // swap
    if (step == 1) {
        if (keep) rvel(i, j, k, nrhs) = r_indx;
        const double cff3 = 0.5 * DC;
        vel(i, j, k) = a - cff3 * r_nrhs + b;
    } else {
        const double cff1 = 5.0 / 12.0, cff2 = 16.0 / 12.0;
        vel(i, j, k) = a + DC * (cff1 * r_indx - cff2 * r_nrhs) + b;
        if (keep) rvel(i, j, k, nrhs) = 0.0;
    }
}
__global__ void k_prestep_uv2(B3 bx, B3 dead_u, B3 dead_v, A4 u, A4 u_old, A4 ru, A4 v, A4 v_old, A4 rv, A4 Hz, A4 sustr, A4 bustr, A4 svstr, A4 bvstr,
                              A4 pm, A4 pn, int step, int nrhs, int N, double dt)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const int k0 = bx.lo2 + RH_KZ * PUV_KC;
    const int indx = nrhs ? 0 : 1;
    double DCu = 0.0, DCv = 0.0;
    if (step != 0) {
        const double pm0 = pm(i, j, 0), pn0 = pn(i, j, 0);
        DCu = 0.25 * dt * (pm0 + pm(i - 1, j, 0)) * (pn0 + pn(i - 1, j, 0));
        DCv = 0.25 * dt * (pm0 + pm(i, j - 1, 0)) * (pn0 + pn(i, j - 1, 0));
    }
    const bool keep_u = !inb2(dead_u, i, j), keep_v = !inb2(dead_v, i, j);
    double su = 0.0, bu = 0.0, sv = 0.0, bv = 0.0;
    if (k0 == 0) { bu = bustr(i, j, 0); bv = bvstr(i, j, 0); }
    if (k0 + PUV_KC > N) { su = sustr(i, j, 0); sv = svstr(i, j, 0); }
#pragma unroll
    for (int kk = 0; kk < PUV_KC; kk++) {
        const int k = k0 + kk;
        if (k <= N) {
            const double hz0 = Hz(i, j, k);
            prestep_uv_point(u, u_old, ru, i, j, k, hz0, Hz(i - 1, j, k), DCu, su, bu, keep_u, step, nrhs, indx, N, dt);
            prestep_uv_point(v, v_old, rv, i, j, k, hz0, Hz(i, j - 1, k), DCv, sv, bv, keep_v, step, nrhs, indx, N, dt);
        }
    }
}
int launch_prestep_uv2(rh_ctx *c, rh_state *s, rh_box bx, rh_box dead_u, rh_box dead_v, int iic, int ntfirst, int nrhs, int N, double dt)
{
    TScope t(c, "prestep_uv");
    bx.lo[2] = 0; bx.hi[2] = N;
    rh_box bsl = bx; bsl.lo[0] -= 1; bsl.lo[1] -= 1;     // (i-1, j) and (i, j-1) of every point
    const rh_box b2 = bx_slab(bx, 2, 0), bs2 = bx_slab(bsl, 2, 0);
    if (s->ru.ncomp < 2 || s->rv.ncomp < 2) return rh_fail("prestep_uv: ru / rv need 2 components");
    RH_NEED(s->xvel_new, bx); RH_NEED(s->xvel_old, bx); RH_NEED(s->ru, bx); RH_NEED(s->yvel_new, bx); RH_NEED(s->yvel_old, bx); RH_NEED(s->rv, bx);
    RH_NEED(s->Hz, bsl); RH_NEED(s->sustr, b2); RH_NEED(s->bustr, b2); RH_NEED(s->svstr, b2); RH_NEED(s->bvstr, b2); RH_NEED(s->pm, bs2); RH_NEED(s->pn, bs2);
    const int step = (iic == ntfirst) ? 0 : (iic == ntfirst + 1 ? 1 : 2);
    dim3 grid = grid2(bx); grid.z = (unsigned)((N + 1 + PUV_KC - 1) / PUV_KC);
    k_prestep_uv2<<<grid, rh_block(), 0, c->stream>>>(mkB(bx), mkB(dead_u), mkB(dead_v), mkA(s->xvel_new), mkA(s->xvel_old), mkA(s->ru),
        mkA(s->yvel_new), mkA(s->yvel_old), mkA(s->rv), mkA(s->Hz), mkA(s->sustr), mkA(s->bustr), mkA(s->svstr), mkA(s->bvstr), mkA(s->pm), mkA(s->pn),
        step, nrhs, N, dt);
    RH_LAUNCH_CHECK(); return 0;
}
static int prestep_uv(rh_ctx *c, rh_box vel_bx, rh_box dead, int ioff, int joff, rh_array4 vel, rh_array4 vel_old, rh_array4 rvel,
                      rh_array4 Hz, rh_array4 sstr, rh_array4 bstr, rh_array4 pm, rh_array4 pn,
                      int iic, int ntfirst, int nrhs, int N, double dt);
extern "C" int rh_prestep_uv(rh_ctx *c, rh_box vel_bx, int ioff, int joff, rh_array4 vel, rh_array4 vel_old, rh_array4 rvel,
                             rh_array4 Hz, rh_array4 sstr, rh_array4 bstr, rh_array4 pm, rh_array4 pn,
                             int iic, int ntfirst, int nrhs, int N, double dt)
{
    return prestep_uv(c, vel_bx, bx_make(0, 0, 0, -1, -1, -1), ioff, joff, vel, vel_old, rvel, Hz, sstr, bstr, pm, pn, iic, ntfirst, nrhs, N, dt);
}
// the call of setup_step: `dead` = the faces on which prsgrd assigns rvel(nrhs) next
int launch_prestep_uv(rh_ctx *c, rh_box vel_bx, rh_box dead, int ioff, int joff, rh_array4 vel, rh_array4 vel_old, rh_array4 rvel,
                      rh_array4 Hz, rh_array4 sstr, rh_array4 bstr, rh_array4 pm, rh_array4 pn, int iic, int ntfirst, int nrhs, int N, double dt)
{
    return prestep_uv(c, vel_bx, dead, ioff, joff, vel, vel_old, rvel, Hz, sstr, bstr, pm, pn, iic, ntfirst, nrhs, N, dt);
}
static int prestep_uv(rh_ctx *c, rh_box vel_bx, rh_box dead, int ioff, int joff, rh_array4 vel, rh_array4 vel_old, rh_array4 rvel,
                      rh_array4 Hz, rh_array4 sstr, rh_array4 bstr, rh_array4 pm, rh_array4 pn,
                      int iic, int ntfirst, int nrhs, int N, double dt)
{
    TScope t(c, "prestep_uv");
    if ((ioff | joff) != 1 || (ioff & joff)) return rh_fail("rh_prestep_uv: (ioff,joff) must be (1,0) or (0,1)");
    vel_bx.lo[2] = 0; vel_bx.hi[2] = N;
    rh_box bs = vel_bx; bs.lo[0] -= ioff; bs.lo[1] -= joff;
    const rh_box b2 = bx_slab(vel_bx, 2, 0), bs2 = bx_slab(bs, 2, 0);
    if (rvel.ncomp < 2) return rh_fail("rh_prestep_uv: rvel needs 2 components");
    RH_NEED(vel, vel_bx); RH_NEED(vel_old, vel_bx); RH_NEED(rvel, vel_bx); RH_NEED(Hz, bs);
    RH_NEED(sstr, b2); RH_NEED(bstr, b2); RH_NEED(pm, bs2); RH_NEED(pn, bs2);
    const int step = (iic == ntfirst) ? 0 : (iic == ntfirst + 1 ? 1 : 2);
    k_prestep_uv<<<grid3(vel_bx), rh_block(), 0, c->stream>>>(mkB(vel_bx), mkB(dead), ioff, joff, mkA(vel), mkA(vel_old), mkA(rvel), mkA(Hz),
        mkA(sstr), mkA(bstr), mkA(pm), mkA(pn), step, nrhs, N, dt);
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::prsgrd, REMORA_prsgrd.cpp:6-197 (density-Jacobian pressure gradient with harmonic-mean slopes).
// Kernel 1: hydrostatic pressure P, one downward sweep per column (:54-99); the vertical slopes dR, dZ are a
// 3-level window in registers instead of two scratch FABs.
__device__ __forceinline__ double hmean_eps(double a, double b)   // :65-67 / :91-96 pattern: 2ab/(a+b) if 2ab > eps else 0
{
    const double eps = 1.0E-10;
    const double cff = 2.0 * a * b;
    return cff > eps ? cff / (a + b) : 0.0;
}
__global__ void k_prsgrd_P(B3 bx, A4 P, A4 rho, A4 z_r, A4 z_w, int N, double GRho, double GRho0, double HalfGRho)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const double OneFifth = 0.2, OneTwelfth = 1.0 / 12.0;
    // raw differences: d(k) = f(k+1)-f(k) for k<N, d(N) = f(N)-f(N-1)
    double rN = rho(i, j, N), rN1 = rho(i, j, N - 1), zN = z_r(i, j, N), zN1 = z_r(i, j, N - 1);
    {
        const double cff1 = 1.0 / (zN - zN1);
        const double zw = z_w(i, j, N + 1);
        const double cff2 = 0.5 * (rN - rN1) * (zw - zN) * cff1;
        P(i, j, N) = GRho0 * zw + GRho * (rN + cff2) * (zw - zN);
    }
    // harmonic slopes at level N: raw d(N) and raw d(N-1) are both f(N)-f(N-1)
    double dRraw_k1 = rN - rN1, dZraw_k1 = zN - zN1;               // raw d(k+1) with k+1 = N
    double dRraw_k = rN - rN1, dZraw_k = zN - zN1;                 // raw d(k) with k = N-1
    double dRh_k1 = hmean_eps(dRraw_k1, dRraw_k);
    double dZh_k1 = 2.0 * dZraw_k1 * dZraw_k / (dZraw_k1 + dZraw_k);
    double r_k1 = rN, z_k1 = zN, r_k = rN1, z_k = zN1, Pk1 = P(i, j, N);
    for (int k = N - 1; k >= 0; k--) {
        // raw d(k-1) (or d(0) itself at k = 0)
        double r_km = 0.0, z_km = 0.0, dRraw_km, dZraw_km;
        if (k > 0) { r_km = rho(i, j, k - 1); z_km = z_r(i, j, k - 1); dRraw_km = r_k - r_km; dZraw_km = z_k - z_km; }
        else { dRraw_km = dRraw_k; dZraw_km = dZraw_k; }
        const double dRh_k = hmean_eps(dRraw_k, dRraw_km);
        const double dZh_k = 2.0 * dZraw_k * dZraw_km / (dZraw_k + dZraw_km);
        const double rho_diff = r_k1 - r_k - OneTwelfth * (dRh_k1 + dRh_k);
        const double z_diff = z_k1 - z_k - OneTwelfth * (dZh_k1 + dZh_k);
        const double rz_avg = (r_k1 + r_k) * (z_k1 - z_k);
        const double Pk = Pk1 + HalfGRho * (rz_avg - OneFifth * ((dRh_k1 - dRh_k) * z_diff - (dZh_k1 - dZh_k) * rho_diff));
        P(i, j, k) = Pk;
        Pk1 = Pk; dRh_k1 = dRh_k; dZh_k1 = dZh_k; r_k1 = r_k; z_k1 = z_k; r_k = r_km; z_k = z_km;
        dRraw_k = dRraw_km; dZraw_k = dZraw_km;
    }
}
// Kernel 2: ru, rv (:103-196), pointwise in k.  The horizontal slopes dRx, dZx are rebuilt from a 4-point window.
__device__ __forceinline__ double hmean_inv(double a, double b)    // :91-96: cff = 2ab; cff > eps ? cff*(1/(a+b)) : 0
{
    const double eps = 1.0E-10;
    const double cff = 2.0 * a * b;
    if (cff > eps) { const double cff1 = 1.0 / (b + a); return cff * cff1; }
    return 0.0;
}
template <class TR, class TZ, class TP, class TH>
__device__ __forceinline__ double prsgrd_point(const TR &rho, const TZ &z_r, const TP &P, const TH &Hz, int i, int j, int k, int di, int dj,
                                               double m1, double m0, double mp, double metric, double HalfGRho)
{   // one momentum point of :103-149 (di = 1) or :152-196 (dj = 1): four-point window along the direction of the gradient
    const double OneFifth = 0.2, OneTwelfth = 1.0 / 12.0;
    const double r2 = rho(i - 2 * di, j - 2 * dj, k), r1 = rho(i - di, j - dj, k), r0 = rho(i, j, k), rp = rho(i + di, j + dj, k);
    const double z2 = z_r(i - 2 * di, j - 2 * dj, k), z1 = z_r(i - di, j - dj, k), z0 = z_r(i, j, k), zp = z_r(i + di, j + dj, k);
    const double FC1 = (r1 - r2) * m1, FC0 = (r0 - r1) * m0, FCp = (rp - r0) * mp;
    const double ax1 = (z1 - z2) * m1, ax0 = (z0 - z1) * m0, axp = (zp - z0) * mp;
    const double dZx0 = hmean_inv(ax0, axp), dZx1 = hmean_inv(ax1, ax0);
    const double dRx0 = hmean_inv(FCp, FC0), dRx1 = hmean_inv(FC0, FC1);
    const double rho_diff = r0 - r1 - OneTwelfth * (dRx0 + dRx1);
    const double z_r_diff = z0 - z1 - OneTwelfth * (dZx0 + dZx1);
    const double Hz_avg = 0.5 * (Hz(i, j, k) + Hz(i - di, j - dj, k));
    return metric * Hz_avg * (P(i - di, j - dj, k) - P(i, j, k) - HalfGRho *
        ((r0 + r1) * (z0 - z1) - OneFifth * ((dRx0 - dRx1) * z_r_diff - (dZx0 - dZx1) * rho_diff)));
}
__global__ void k_prsgrd_uv(B3 ub, B3 vb, B3 un, A4 ru, A4 rv, A4 pn, A4 pm, A4 rho, A4 P, A4 Hz, A4 z_r, A4 msku, A4 mskv,
                            int nrhs, double HalfGRho)
{
    RH_IJ(un);
    const int k = un.lo2 + RH_KZ;
    if (inb2(ub, i, j)) {
        const double on_u = 2.0 / (pn(i - 1, j, 0) + pn(i, j, 0));
        ru(i, j, k, nrhs) = prsgrd_point(rho, z_r, P, Hz, i, j, k, 1, 0, msku(i - 1, j, 0), msku(i, j, 0), msku(i + 1, j, 0), on_u, HalfGRho);
    }
    if (inb2(vb, i, j)) {
        const double om_v = 2.0 / (pm(i, j - 1, 0) + pm(i, j, 0));
        rv(i, j, k, nrhs) = prsgrd_point(rho, z_r, P, Hz, i, j, k, 0, 1, mskv(i, j - 1, 0), mskv(i, j, 0), mskv(i, j + 1, 0), om_v, HalfGRho);
    }
}
// LDS-tiled form: the direct form asks the fabric for 13 doubles per cell where 4 are distinct (profiles/r1_v3_traffic.json).  A
// block walks the column of one 64 x 8 tile: per level it stages rho, z_r (halo 2) and P, Hz (halo 1) in LDS (double-buffered on
// the parity of k, the next level's values already on their way), and every stencil neighbour is read from LDS.
// rho_eos and the hydrostatic pressure of prsgrd in ONE downward sweep (setup_step calls them back to back on unchanged operands:
// REMORA_setup_step.cpp:200 and :229-260).  rho(k) is formed from T, S as in k_rho_eos and P(k+1) follows one level behind with the
// three densities it needs still in registers, so prsgrd never re-reads rho and z_r.  Same expressions, same order: bit-identical
// to k_rho_eos followed by k_prsgrd_P.  P is written on pbx (the grow-1 box) only.
__global__ void k_rho_eos_P(B3 bx, B3 pbx, A4 st, A4 rho, A4 rhoA, A4 rhoS, A4 bvf, A4 Hz, A4 z_w, A4 z_r, A4 h, A4 mskr, A4 P, int N,
                            double R0, double S0, double T0, double Tcoef, double Scoef, double rho0, double g)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const bool doP = inb2(pbx, i, j);
    const double m = mskr(i, j, 0);
    const double gorho0 = g / rho0;
    const double GRho = g / rho0, GRho0 = 1000.0 * GRho, HalfGRho = 0.5 * GRho;
    const double OneFifth = 0.2, OneTwelfth = 1.0 / 12.0;
    const double zwtop = z_w(i, j, N + 1);
    double r_up = 0.0, z_up = 0.0, A = 0.0, S = 0.0;
    double dRraw_k = 0, dZraw_k = 0, dRh_k1 = 0, dZh_k1 = 0, r_k1 = 0, z_k1 = 0, r_k = 0, z_k = 0, Pk1 = 0;
    for (int k = N; k >= 0; k--) {
        const double r = (R0 - R0 * Tcoef * (st(i, j, k, 0) - T0) + R0 * Scoef * (st(i, j, k, 1) - S0) - 1000.0) * m;
        rho(i, j, k) = r;
        const double hz = Hz(i, j, k), zr = z_r(i, j, k);
        if (k == N) { const double cff0 = r * hz; S = 0.5 * cff0 * hz; A = cff0; }
        else {
            const double cff1 = r * hz; S += hz * (A + 0.5 * cff1); A += cff1;
            bvf(i, j, k + 1) = -gorho0 * (r_up - r) / (z_up - zr);
        }
        if (doP) {
            if (k == N - 1) {                              // P(N) and the slopes at level N, prsgrd.cpp:54-79
                const double rN = r_up, rN1 = r, zN = z_up, zN1 = zr;
                const double cff1 = 1.0 / (zN - zN1);
                const double cff2 = 0.5 * (rN - rN1) * (zwtop - zN) * cff1;
                Pk1 = GRho0 * zwtop + GRho * (rN + cff2) * (zwtop - zN);
                P(i, j, N) = Pk1;
                const double dRraw_k1 = rN - rN1, dZraw_k1 = zN - zN1;
                dRraw_k = rN - rN1; dZraw_k = zN - zN1;
                dRh_k1 = hmean_eps(dRraw_k1, dRraw_k);
                dZh_k1 = 2.0 * dZraw_k1 * dZraw_k / (dZraw_k1 + dZraw_k);
                r_k1 = rN; z_k1 = zN; r_k = rN1; z_k = zN1;
            } else if (k <= N - 2) {                       // P(k+1): r, zr are level k = (k+1) - 1
                const double r_km = r, z_km = zr;
                const double dRraw_km = r_k - r_km, dZraw_km = z_k - z_km;
                const double dRh_k = hmean_eps(dRraw_k, dRraw_km);
                const double dZh_k = 2.0 * dZraw_k * dZraw_km / (dZraw_k + dZraw_km);
                const double rho_diff = r_k1 - r_k - OneTwelfth * (dRh_k1 + dRh_k);
                const double z_diff = z_k1 - z_k - OneTwelfth * (dZh_k1 + dZh_k);
                const double rz_avg = (r_k1 + r_k) * (z_k1 - z_k);
                const double Pk = Pk1 + HalfGRho * (rz_avg - OneFifth * ((dRh_k1 - dRh_k) * z_diff - (dZh_k1 - dZh_k) * rho_diff));
                P(i, j, k + 1) = Pk;
                Pk1 = Pk; dRh_k1 = dRh_k; dZh_k1 = dZh_k; r_k1 = r_k; z_k1 = z_k; r_k = r_km; z_k = z_km;
                dRraw_k = dRraw_km; dZraw_k = dZraw_km;
            }
        }
        r_up = r; z_up = zr;
    }
    if (doP) {                                             // P(0): the slope below level 0 repeats d(0)
        const double dRraw_km = dRraw_k, dZraw_km = dZraw_k;
        const double dRh_k = hmean_eps(dRraw_k, dRraw_km);
        const double dZh_k = 2.0 * dZraw_k * dZraw_km / (dZraw_k + dZraw_km);
        const double rho_diff = r_k1 - r_k - OneTwelfth * (dRh_k1 + dRh_k);
        const double z_diff = z_k1 - z_k - OneTwelfth * (dZh_k1 + dZh_k);
        const double rz_avg = (r_k1 + r_k) * (z_k1 - z_k);
        P(i, j, 0) = Pk1 + HalfGRho * (rz_avg - OneFifth * ((dRh_k1 - dRh_k) * z_diff - (dZh_k1 - dZh_k) * rho_diff));
    }
    const double cff2 = 1.0 / rho0;
    const double cff11 = 1.0 / (zwtop + h(i, j, 0, 0));
    A *= cff2 * cff11;
    S *= 2.0 * cff11 * cff11 * cff2;
    rhoA(i, j, 0) = A; rhoS(i, j, 0) = S;
}
// setup_step's form of rho_eos: also leaves prsgrd's P behind (c->P_ready tells the following rh_prsgrd to skip its first kernel)
int launch_rho_eos_P(rh_ctx *c, rh_state *s, rh_box bx, rh_box pbx)
{
    TScope t(c, "rho_eos");
    const int N = c->N;
    const rh_box b2 = bx_slab(bx, 2, 0);
    rh_box bw = bx; bw.lo[2] = 0; bw.hi[2] = N + 1;
    bx.lo[2] = 0; bx.hi[2] = N; pbx.lo[2] = 0; pbx.hi[2] = N;
    if (s->cons_old.ncomp < 2) return rh_fail("rho_eos: state needs temp and salt");
    if (!bx_contains(bx, pbx)) return rh_fail("rho_eos: pressure box outside the density box");
    RH_NEED(s->cons_old, bx); RH_NEED(c->rho, bx); RH_NEED(s->Hz, bx); RH_NEED(s->z_r, bx); RH_NEED(s->z_w, bw); RH_NEED(s->bvf, bw);
    RH_NEED(s->rhoA, b2); RH_NEED(s->rhoS, b2); RH_NEED(s->h, b2); RH_NEED(s->mskr, b2); RH_NEED(c->P, pbx);
    const rh_params &P = c->prm;
    k_rho_eos_P<<<grid2(bx), rh_block(), 0, c->stream>>>(mkB(bx), mkB(pbx), mkA(s->cons_old), mkA(c->rho), mkA(s->rhoA), mkA(s->rhoS), mkA(s->bvf), mkA(s->Hz),
        mkA(s->z_w), mkA(s->z_r), mkA(s->h), mkA(s->mskr), mkA(c->P), N, P.R0, P.S0, P.T0, P.Tcoef, P.Scoef, P.rho0, P.g);
    RH_LAUNCH_CHECK();
    c->P_ready = true;
    return 0;
}

#define PG_BY 16        // rows per block: halo rows are (PG_BY + 2 HY) / PG_BY of the tile reads
__global__ void __launch_bounds__(RT_BX * PG_BY)
k_prsgrd_uv_tiled(B3 ub, B3 vb, B3 un, A4 ru, A4 rv, A4 pn, A4 pm, A4 rho, A4 P, A4 Hz, A4 z_r, A4 msku, A4 mskv, int nrhs, double HalfGRho)
{
    typedef Tile<2, 2, PG_BY> T2; typedef Tile<1, 1, PG_BY> T1;
    constexpr int SET = 2 * T2::SIZE + 2 * T1::SIZE;
    __shared__ double lds[2 * SET];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const RhTile rh_t_ = rh_tile();
    const int i0 = un.lo0 + rh_t_.x * RT_BX, j0 = un.lo1 + rh_t_.y * PG_BY;
    const int i = i0 + tx, j = j0 + ty;
    const bool au = inb2(ub, i, j), av = inb2(vb, i, j);
    double on_u = 0.0, om_v = 0.0, mu1 = 0.0, mu0 = 0.0, mup = 0.0, mv1 = 0.0, mv0 = 0.0, mvp = 0.0;
    if (au) { on_u = 2.0 / (pn(i - 1, j, 0) + pn(i, j, 0)); mu1 = msku(i - 1, j, 0); mu0 = msku(i, j, 0); mup = msku(i + 1, j, 0); }
    if (av) { om_v = 2.0 / (pm(i, j - 1, 0) + pm(i, j, 0)); mv1 = mskv(i, j - 1, 0); mv0 = mskv(i, j, 0); mvp = mskv(i, j + 1, 0); }
    const HaloCell hr = halo_cell<2, 2, PG_BY>(rho, i0, j0, tx, ty), hz = halo_cell<2, 2, PG_BY>(z_r, i0, j0, tx, ty),
                   hp = halo_cell<1, 1, PG_BY>(P, i0, j0, tx, ty), hh = halo_cell<1, 1, PG_BY>(Hz, i0, j0, tx, ty);
    const int k0 = un.lo2, k1 = un.hi2;
    double r0 = ld_in(rho, i, j, k0), z0 = ld_in(z_r, i, j, k0), p0 = ld_in(P, i, j, k0), h0 = ld_in(Hz, i, j, k0);
    double rh = halo_ld(rho, hr, k0), zh = halo_ld(z_r, hz, k0), ph = halo_ld(P, hp, k0), hhv = halo_ld(Hz, hh, k0);
    for (int k = k0; k <= k1; k++) {
        double *set = lds + ((k - k0) & 1) * SET;
        double *sr = set, *sz = sr + T2::SIZE, *sp = sz + T2::SIZE, *sh = sp + T1::SIZE;
        tile_put<2, 2>(sr, tx, ty, r0, hr, rh); tile_put<2, 2>(sz, tx, ty, z0, hz, zh);
        tile_put<1, 1>(sp, tx, ty, p0, hp, ph); tile_put<1, 1>(sh, tx, ty, h0, hh, hhv);
        if (k + 1 <= k1) {
            r0 = ld_in(rho, i, j, k + 1); z0 = ld_in(z_r, i, j, k + 1); p0 = ld_in(P, i, j, k + 1); h0 = ld_in(Hz, i, j, k + 1);
            rh = halo_ld(rho, hr, k + 1); zh = halo_ld(z_r, hz, k + 1); ph = halo_ld(P, hp, k + 1); hhv = halo_ld(Hz, hh, k + 1);
        }
        __syncthreads();
        const TView<2, 2, PG_BY> vr{sr, i0, j0}, vz{sz, i0, j0}; const TView<1, 1, PG_BY> vp{sp, i0, j0}, vh{sh, i0, j0};
        if (au) ru(i, j, k, nrhs) = prsgrd_point(vr, vz, vp, vh, i, j, k, 1, 0, mu1, mu0, mup, on_u, HalfGRho);
        if (av) rv(i, j, k, nrhs) = prsgrd_point(vr, vz, vp, vh, i, j, k, 0, 1, mv1, mv0, mvp, om_v, HalfGRho);
    }
}
// ---- rho_eos + the hydrostatic pressure + the horizontal differences of prsgrd in ONE downward march (setup_step.cpp:200 and :262-266:
// nothing between the two calls touches their operands).  k_rho_eos_P + k_prsgrd_uv_tiled hand rho and P over through HBM (written
// once, read 1.2-1.3 times with the halo rows) although both are column-local functions of T, S, Hz, z_r.  Here a block owns a
// 64 x 12 tile of momentum points and marches EVERY column of the tile grown by the stencil of a momentum point -- two cells below,
// one above (67 x 15 = 1005 columns, one per thread of the block's 16 wavefronts; see RPTile).  Per level a column forms rho(k) from T, S (rho_eos.cpp:54-59),
// the column sums and bvf (:69-98) and prsgrd's P(k+1) one level behind (prsgrd.cpp:54-99) -- the body of k_rho_eos_P -- and puts
// rho, z_r, Hz of level k (ring of three levels) and P(k+1) (two) into LDS; after the barrier the tile's threads form ru, rv of
// level k+1 from LDS with k_prsgrd_uv_tiled's own expression.  rho and P never reach memory: 14 -> ~8 doubles per cell for the two
// leaves.  Halo columns are marched once per sharing block: the same operands through the same expressions, hence the same bits.
// bvf, rhoA, rhoS (wanted on the cell box grown by two, REMORA_setup_step.cpp:166-200) are stored by the block that owns the
// point: the block whose tile holds the point clamped into the box of momentum points.
#define RP_BY 12       // rows of momentum points per block
#define RP_WAVES 16    // wavefronts per block: RP_BY own the tile, the rest march the halo ring
// what a marching column carries from level to level.  k_rho_eos_P's r_up / z_up are r_k / z_k here and its raw slopes dRraw_k / dZraw_k
// are r_k1 - r_k / z_k1 - z_k, formed again where they are used: the same operands through the same operation, hence the same bits,
// and four doubles fewer per column in a kernel whose threads carry two columns within 128 registers.
struct RPState { double A, S, dRh_k1, dZh_k1, r_k1, z_k1, r_k, z_k, Pk1; };
struct RPConst { double R0, S0, T0, Tcoef, Scoef, rho0, g; };
// level k of one column, top-down; returns rho(k), Pout = P(k+1) for k <= N-1 (the statements of k_rho_eos_P); zwtop is only read at k = N-1
template <class ZW>
__device__ __forceinline__ double rp_level(RPState &q, int k, int N, double t, double sa, double hz, double zr, double m, const ZW &zwtop,
                                           bool own, const A4 &bvf, int gi, int gj, const RPConst &c, double &Pout)
{
    const double gorho0 = c.g / c.rho0;
    const double GRho = c.g / c.rho0, GRho0 = 1000.0 * GRho, HalfGRho = 0.5 * GRho;
    const double OneFifth = 0.2, OneTwelfth = 1.0 / 12.0;
    const double r = (c.R0 - c.R0 * c.Tcoef * (t - c.T0) + c.R0 * c.Scoef * (sa - c.S0) - 1000.0) * m;
    if (k == N) { const double cff0 = r * hz; q.S = 0.5 * cff0 * hz; q.A = cff0; q.r_k = r; q.z_k = zr; }
    else {
        const double cff1 = r * hz; q.S += hz * (q.A + 0.5 * cff1); q.A += cff1;
        const double b = -gorho0 * (q.r_k - r) / (q.z_k - zr);               // r_k, z_k: the level above (r_up, z_up)
        if (own) bvf(gi, gj, k + 1) = b;
        if (k == N - 1) {
            const double rN = q.r_k, rN1 = r, zN = q.z_k, zN1 = zr, zw = zwtop();
            const double cff1p = 1.0 / (zN - zN1);
            const double cff2 = 0.5 * (rN - rN1) * (zw - zN) * cff1p;
            q.Pk1 = GRho0 * zw + GRho * (rN + cff2) * (zw - zN);
            Pout = q.Pk1;
            const double dRraw_k1 = rN - rN1, dZraw_k1 = zN - zN1;
            const double dRraw_k = rN - rN1, dZraw_k = zN - zN1;
            q.dRh_k1 = hmean_eps(dRraw_k1, dRraw_k);
            q.dZh_k1 = 2.0 * dZraw_k1 * dZraw_k / (dZraw_k1 + dZraw_k);
            q.r_k1 = rN; q.z_k1 = zN; q.r_k = rN1; q.z_k = zN1;
        } else {
            const double r_km = r, z_km = zr;
            const double dRraw_k = q.r_k1 - q.r_k, dZraw_k = q.z_k1 - q.z_k;
            const double dRraw_km = q.r_k - r_km, dZraw_km = q.z_k - z_km;
            const double dRh_k = hmean_eps(dRraw_k, dRraw_km);
            const double dZh_k = 2.0 * dZraw_k * dZraw_km / (dZraw_k + dZraw_km);
            const double rho_diff = q.r_k1 - q.r_k - OneTwelfth * (q.dRh_k1 + dRh_k);
            const double z_diff = q.z_k1 - q.z_k - OneTwelfth * (q.dZh_k1 + dZh_k);
            const double rz_avg = (q.r_k1 + q.r_k) * (q.z_k1 - q.z_k);
            const double Pk = q.Pk1 + HalfGRho * (rz_avg - OneFifth * ((q.dRh_k1 - dRh_k) * z_diff - (q.dZh_k1 - dZh_k) * rho_diff));
            Pout = Pk;
            q.Pk1 = Pk; q.dRh_k1 = dRh_k; q.dZh_k1 = dZh_k; q.r_k1 = q.r_k; q.z_k1 = q.z_k; q.r_k = r_km; q.z_k = z_km;
        }
    }
    return r;
}
__device__ __forceinline__ double rp_P0(const RPState &q, const RPConst &c)      // P(0): the slope below level 0 repeats d(0)
{
    const double HalfGRho = 0.5 * (c.g / c.rho0), OneFifth = 0.2, OneTwelfth = 1.0 / 12.0;
    const double dRraw_k = q.r_k1 - q.r_k, dZraw_k = q.z_k1 - q.z_k;
    const double dRraw_km = dRraw_k, dZraw_km = dZraw_k;
    const double dRh_k = hmean_eps(dRraw_k, dRraw_km);
    const double dZh_k = 2.0 * dZraw_k * dZraw_km / (dZraw_k + dZraw_km);
    const double rho_diff = q.r_k1 - q.r_k - OneTwelfth * (q.dRh_k1 + dRh_k);
    const double z_diff = q.z_k1 - q.z_k - OneTwelfth * (q.dZh_k1 + dZh_k);
    const double rz_avg = (q.r_k1 + q.r_k) * (q.z_k1 - q.z_k);
    return q.Pk1 + HalfGRho * (rz_avg - OneFifth * ((q.dRh_k1 - dRh_k) * z_diff - (q.dZh_k1 - dZh_k) * rho_diff));
}
// Block = 16 wavefronts: RP_BY = 12 of them own the tile's 64 x 12 momentum points (and march those columns), the other four march
// the 237 columns of the halo ring -- ONE column per thread (two per thread did not fit 128 registers: 49 spilled), 1005 of 1024
// threads busy in the march, the twelve tile waves alone in the momentum stage.  The ring is as wide as the stencil of a momentum
// point: two cells below (i-2, j-2), ONE above (i+1, j+1) -- with two on every side (68 x 15 columns for 64 x 11 points) every output
// point cost 1.45 evaluations of the equation of state, now 1.31.
struct RPTile {
    static constexpr int PX = RT_BX + 3, PY = RP_BY + 3, SIZE = PX * PY, NROWCELLS = 3 * PX, NHALO = NROWCELLS + 3 * RP_BY;
};
struct RPView { const double *s; int i0, j0; __device__ __forceinline__ double operator()(int i, int j, int) const { return s[(j - j0 + 2) * RPTile::PX + (i - i0 + 2)]; } };
__global__ void __launch_bounds__(RT_BX * RP_WAVES)
k_rho_prsgrd(B3 gb, B3 ub, B3 vb, B3 un, A4 st, A4 rhoA, A4 rhoS, A4 bvf, A4 Hz, A4 z_w, A4 z_r, A4 h, A4 mskr,
             A4 ru, A4 rv, A4 pn, A4 pm, A4 msku, A4 mskv, int nrhs, int N, RPConst cst)
{
    typedef RPTile T2;
    constexpr int SZ = T2::SIZE + 4;                        // + a spare slot: the four threads without a column park their values there
    static_assert(T2::NHALO <= RT_BX * (RP_WAVES - RP_BY), "one halo column per thread of the halo waves");
    __shared__ double lds[13 * SZ];                         // rho, z_r, Hz: three levels each; P: two; msku, mskv
    double *const sR = lds, *const sZ = lds + 3 * SZ, *const sH = lds + 6 * SZ, *const sP = lds + 9 * SZ;
    double *const sMu = lds + 11 * SZ, *const sMv = lds + 12 * SZ;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const RhTile rh_t_ = rh_tile_patch<8, 4>();          // one block per CU: the 32 resident on an XCD form a compact patch and share halo columns in its L2
    const int i0 = un.lo0 + rh_t_.x * RT_BX, j0 = un.lo1 + rh_t_.y * RP_BY;
    const bool tile_wave = ty < RP_BY;                      // wave-uniform
    // this thread's column: its own point (tile waves) or one cell of the halo ring (halo waves)
    int x, y;                                               // position inside the grown tile
    bool has = true;
    if (tile_wave) { x = tx + 2; y = ty + 2; }
    else {
        const int t = (ty - RP_BY) * RT_BX + tx;
        has = t < T2::NHALO;
        x = 0; y = 0;
        if (has) {
            if (t < T2::NROWCELLS) { const int r = t / T2::PX; x = t - r * T2::PX; y = r < 2 ? r : RP_BY + 2; }
            else { const int u = t - T2::NROWCELLS; const int r = u / 3; const int xx = u - r * 3; y = 2 + r; x = xx < 2 ? xx : RT_BX + 2; }
        }
    }
    const int gi = i0 - 2 + x, gj = j0 - 2 + y;             // the column's cell
    // every thread marches: no branch around the loads (at a join of paths with different numbers of loads in flight hipcc waits for
    // ALL of them, vmcnt(0), and the two-level prefetch below would be drained at every level)
    const int l0 = has ? y * T2::PX + x : T2::SIZE;
    const double HalfGRho = 0.5 * (cst.g / cst.rho0);
    const bool au = tile_wave && inb2(ub, gi, gj), av = tile_wave && inb2(vb, gi, gj);
    double on_u = 0.0, om_v = 0.0;
    if (au) on_u = 2.0 / (pn(gi - 1, gj, 0) + pn(gi, gj, 0));
    if (av) om_v = 2.0 / (pm(gi, gj - 1, 0) + pm(gi, gj, 0));
    const bool act = has && inb2(gb, gi, gj);
    // loads of a column that is not wanted go to a point that exists (its values are never used)
    const int ci = min(max(gi, gb.lo0), gb.hi0), cj = min(max(gj, gb.lo1), gb.hi1);
    bool own;
    {   // the block that stores bvf / rhoA / rhoS of a cell: the one whose tile holds the cell clamped into the box of momentum points
        const int qi = min(max(gi, un.lo0), un.hi0), qj = min(max(gj, un.lo1), un.hi1);
        own = act && qi >= i0 && qi < i0 + RT_BX && qj >= j0 && qj < j0 + RP_BY;
    }
    // the velocity masks of the tile, staged once (the momentum points read them at i-1 .. i+1 / j-1 .. j+1 at every level)
    auto m_in = [&](const A4 &a) -> double { return (gi >= a.lo0 && gi <= a.hi0 && gj >= a.lo1 && gj <= a.hi1) ? a(gi, gj, 0) : 0.0; };
    sMu[l0] = m_in(msku); sMv[l0] = m_in(mskv);
    const double m0 = mskr(ci, cj, 0);
    const unsigned oW = z_w.ix(ci, cj, N + 1);
    auto zw = [&]() -> double { return z_w.p[oW]; };
    RPState q;
    q.A = q.S = q.dRh_k1 = q.dZh_k1 = q.r_k1 = q.z_k1 = q.r_k = q.z_k = q.Pk1 = 0.0;
    // operands of a level: requested TWO levels ahead (two register sets, the loop unrolled by two so that no copy ties a set to the
    // wait of the other: the wave waits for the older level only).  One level ahead left the waves parked on memory two thirds of
    // their cycles (PMC, profiles/r3_*_sq_counters.txt): the momentum stage of one level is shorter than a loaded memory round trip.
    struct Lvl { double t, s, h, z; };
    const unsigned oT = st.ix(ci, cj, 0), oH = Hz.ix(ci, cj, 0), oZ = z_r.ix(ci, cj, 0);
    auto load = [&](int k) -> Lvl {
        const unsigned kk = (unsigned)(k > 0 ? k : 0);
        Lvl v; v.t = st.p[oT + kk * st.sz]; v.s = st.p[oT + kk * st.sz + st.sn]; v.h = Hz.p[oH + kk * Hz.sz]; v.z = z_r.p[oZ + kk * z_r.sz]; return v;
    };
    auto momentum = [&](int kk) {                            // ru, rv of level kk from the staged tiles (k_prsgrd_uv_tiled)
        const int r3 = kk % 3;
        const RPView vr{sR + r3 * SZ, i0, j0}, vz{sZ + r3 * SZ, i0, j0}, vh{sH + r3 * SZ, i0, j0}, vp{sP + (kk & 1) * SZ, i0, j0};
        if (au) ru(gi, gj, kk, nrhs) = prsgrd_point(vr, vz, vp, vh, gi, gj, kk, 1, 0, sMu[l0 - 1], sMu[l0], sMu[l0 + 1], on_u, HalfGRho);
        if (av) rv(gi, gj, kk, nrhs) = prsgrd_point(vr, vz, vp, vh, gi, gj, kk, 0, 1, sMv[l0 - T2::PX], sMv[l0], sMv[l0 + T2::PX], om_v, HalfGRho);
    };
    auto level = [&](int k, const Lvl &v) {
        const int r3 = k % 3;
        double Pn = 0.0;
        const double ra = rp_level(q, k, N, v.t, v.s, v.h, v.z, m0, zw, own, bvf, gi, gj, cst, Pn);
        sR[r3 * SZ + l0] = ra; sZ[r3 * SZ + l0] = v.z; sH[r3 * SZ + l0] = v.h; if (k < N) sP[((k + 1) & 1) * SZ + l0] = Pn;
    };
    Lvl a = load(N), b = load(N - 1);
    int k = N;
    for (; k - 1 >= 0; k -= 2) {                            // two levels per trip, no test on the second (it would make the wait counts conservative)
        level(k, a);
        a = load(k - 2);
        __syncthreads();
        if (tile_wave && k < N) momentum(k + 1);
        level(k - 1, b);
        b = load(k - 3);
        __syncthreads();
        if (tile_wave) momentum(k);
    }
    if (k >= 0) {                                           // an odd number of levels: the last one
        level(k, a);
        __syncthreads();
        if (tile_wave && k < N) momentum(k + 1);
    }
    sP[l0] = rp_P0(q, cst);
    __syncthreads();
    if (tile_wave) momentum(0);
    if (own) {
        const double cff2 = 1.0 / cst.rho0;
        const double cff11 = 1.0 / (zw() + h(gi, gj, 0, 0));
        rhoA(gi, gj, 0) = q.A * (cff2 * cff11); rhoS(gi, gj, 0) = q.S * (2.0 * cff11 * cff11 * cff2);
    }
}
// setup_step's rho_eos (:200) and prsgrd (:262-266) as the one launch above; *applied = 0 when the form does not apply
int launch_rho_prsgrd(rh_ctx *c, rh_state *s, rh_box gbx2, rh_box xbx, rh_box ybx, int nrhs, int N, int *applied)
{
    *applied = 0;
    if (c->unfused || c->no_rhofuse) return 0;
    TScope t(c, "rho_prsgrd");
    gbx2.lo[2] = 0; gbx2.hi[2] = N; xbx.lo[2] = 0; xbx.hi[2] = N; ybx.lo[2] = 0; ybx.hi[2] = N;
    const rh_box b2 = bx_slab(gbx2, 2, 0);
    rh_box bw = gbx2; bw.hi[2] = N + 1;
    const rh_box un = bx_union(xbx, ybx);
    // every cell the momentum points read (i-2 .. i+1, j-2 .. j+1) must be a column this launch marches
    { rh_box need = un; need.lo[0] -= 2; need.lo[1] -= 2; need.hi[0] += 1; need.hi[1] += 1;
      rh_box nu = xbx; nu.lo[0] -= 2; nu.hi[0] += 1; rh_box nv = ybx; nv.lo[1] -= 2; nv.hi[1] += 1;
      if (!bx_contains(gbx2, nu) || !bx_contains(gbx2, nv)) return 0; (void)need; }
    if (s->cons_old.ncomp < 2) return rh_fail("rho_eos: state needs temp and salt");
    RH_NEED(s->cons_old, gbx2); RH_NEED(s->Hz, gbx2); RH_NEED(s->z_r, gbx2); RH_NEED(s->z_w, bw); RH_NEED(s->bvf, bw);
    RH_NEED(s->rhoA, b2); RH_NEED(s->rhoS, b2); RH_NEED(s->h, b2); RH_NEED(s->mskr, b2);
    { rh_box q = xbx; q.lo[0] -= 1; RH_NEED(s->pn, bx_slab(q, 2, 0)); q.hi[0] += 1; RH_NEED(s->msku, bx_slab(q, 2, 0)); }
    { rh_box q = ybx; q.lo[1] -= 1; RH_NEED(s->pm, bx_slab(q, 2, 0)); q.hi[1] += 1; RH_NEED(s->mskv, bx_slab(q, 2, 0)); }
    RH_NEED(s->ru, xbx); RH_NEED(s->rv, ybx);
    if (nrhs < 0 || nrhs >= s->ru.ncomp || nrhs >= s->rv.ncomp) return rh_fail("rho_prsgrd: nrhs out of range");
    const rh_params &P = c->prm;
    const RPConst cst{P.R0, P.S0, P.T0, P.Tcoef, P.Scoef, P.rho0, P.g};
    const dim3 pg((un.hi[0] - un.lo[0] + RT_BX) / RT_BX, (un.hi[1] - un.lo[1] + RP_BY) / RP_BY, 1);
    k_rho_prsgrd<<<pg, dim3(RT_BX, RP_WAVES, 1), 0, c->stream>>>(mkB(gbx2), mkB(xbx), mkB(ybx), mkB(un), mkA(s->cons_old), mkA(s->rhoA), mkA(s->rhoS),
        mkA(s->bvf), mkA(s->Hz), mkA(s->z_w), mkA(s->z_r), mkA(s->h), mkA(s->mskr), mkA(s->ru), mkA(s->rv), mkA(s->pn), mkA(s->pm), mkA(s->msku),
        mkA(s->mskv), nrhs, N, cst);
    RH_LAUNCH_CHECK();
    *applied = 1;
    return 0;
}
extern "C" int rh_prsgrd(rh_ctx *c, rh_box phi_bx, rh_box utbx, rh_box vtbx, rh_array4 ru, rh_array4 rv, rh_array4 pn, rh_array4 pm,
                         rh_array4 rho, rh_array4 P, rh_array4 Hz, rh_array4 z_r, rh_array4 z_w, rh_array4 msku, rh_array4 mskv,
                         int nrhs, int N)
{
    TScope t(c, "prsgrd");
    phi_bx.lo[2] = 0; phi_bx.hi[2] = N; utbx.lo[2] = 0; utbx.hi[2] = N; vtbx.lo[2] = 0; vtbx.hi[2] = N;
    rh_box bw = phi_bx; bw.lo[2] = bw.hi[2] = N + 1;
    RH_NEED(P, phi_bx); RH_NEED(rho, phi_bx); RH_NEED(z_r, phi_bx); RH_NEED(z_w, bw);
    const rh_params &Pm = c->prm;
    const double GRho = Pm.g / Pm.rho0, GRho0 = 1000.0 * GRho, HalfGRho = 0.5 * GRho;
    if (c->P_ready && P.p == c->P.p) c->P_ready = false;          // left behind by launch_rho_eos_P on this very box
    else
    k_prsgrd_P<<<grid2(phi_bx), rh_block(), 0, c->stream>>>(mkB(phi_bx), mkA(P), mkA(rho), mkA(z_r), mkA(z_w), N, GRho, GRho0, HalfGRho);
    RH_LAUNCH_CHECK();
    rh_box us = utbx; us.lo[0] -= 2; us.hi[0] += 1; rh_box vs = vtbx; vs.lo[1] -= 2; vs.hi[1] += 1;
    RH_NEED(rho, us); RH_NEED(rho, vs); RH_NEED(z_r, us); RH_NEED(z_r, vs);
    { rh_box q = utbx; q.lo[0] -= 1; RH_NEED(P, q); RH_NEED(Hz, q); RH_NEED(pn, bx_slab(q, 2, 0)); q.hi[0] += 1; RH_NEED(msku, bx_slab(q, 2, 0)); }
    { rh_box q = vtbx; q.lo[1] -= 1; RH_NEED(P, q); RH_NEED(Hz, q); RH_NEED(pm, bx_slab(q, 2, 0)); q.hi[1] += 1; RH_NEED(mskv, bx_slab(q, 2, 0)); }
    RH_NEED(ru, utbx); RH_NEED(rv, vtbx);
    if (nrhs < 0 || nrhs >= ru.ncomp || nrhs >= rv.ncomp) return rh_fail("rh_prsgrd: nrhs out of range");
    const rh_box un = bx_union(utbx, vtbx);
    if (c->unfused)
        k_prsgrd_uv<<<grid3(un), rh_block(), 0, c->stream>>>(mkB(utbx), mkB(vtbx), mkB(un), mkA(ru), mkA(rv), mkA(pn), mkA(pm), mkA(rho),
            mkA(P), mkA(Hz), mkA(z_r), mkA(msku), mkA(mskv), nrhs, HalfGRho);
    else {
        const dim3 pg((un.hi[0] - un.lo[0] + RT_BX) / RT_BX, (un.hi[1] - un.lo[1] + PG_BY) / PG_BY, 1);
        k_prsgrd_uv_tiled<<<pg, dim3(RT_BX, PG_BY, 1), 0, c->stream>>>(mkB(utbx), mkB(vtbx), mkB(un), mkA(ru), mkA(rv), mkA(pn), mkA(pm), mkA(rho),
            mkA(P), mkA(Hz), mkA(z_r), mkA(msku), mkA(mskv), nrhs, HalfGRho);
    }
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::t3dmix, REMORA_t3dmix.cpp:6-58: harmonic tracer diffusion along s-surfaces, one component per launch.
// d2 carries the reference's coefficient lookup: diff2(i,j,n) is a 3-index Array4 access, i.e. level k = n of
// component 0 (:34,43), so the caller passes the (k, component) pair that the reference actually reads.
__global__ void k_t3dmix(B3 bx, A4 st, A4 rhs, A4 d2, int d2k, A4 Hz, A4 pm, A4 pn, A4 msku, A4 mskv, double dt)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const int k = bx.lo2 + RH_KZ;
    const double r0 = rhs(i, j, k), hz0 = Hz(i, j, k);
    double FX0, FX1, FE0, FE1;
    {
        const double pmon_u = (pm(i - 1, j, 0) + pm(i, j, 0)) / (pn(i - 1, j, 0) + pn(i, j, 0));
        const double cff = 0.25 * (d2(i, j, d2k) + d2(i - 1, j, d2k)) * pmon_u;
        FX0 = cff * (hz0 + Hz(i - 1, j, k)) * (r0 - rhs(i - 1, j, k)); FX0 *= msku(i, j, 0);
    }
    {
        const double pmon_u = (pm(i, j, 0) + pm(i + 1, j, 0)) / (pn(i, j, 0) + pn(i + 1, j, 0));
        const double cff = 0.25 * (d2(i + 1, j, d2k) + d2(i, j, d2k)) * pmon_u;
        FX1 = cff * (Hz(i + 1, j, k) + hz0) * (rhs(i + 1, j, k) - r0); FX1 *= msku(i + 1, j, 0);
    }
    {
        const double pnom_v = (pn(i, j - 1, 0) + pn(i, j, 0)) / (pm(i, j - 1, 0) + pm(i, j, 0));
        const double cff = 0.25 * (d2(i, j, d2k) + d2(i, j - 1, d2k)) * pnom_v;
        FE0 = cff * (hz0 + Hz(i, j - 1, k)) * (r0 - rhs(i, j - 1, k)); FE0 *= mskv(i, j, 0);
    }
    {
        const double pnom_v = (pn(i, j, 0) + pn(i, j + 1, 0)) / (pm(i, j, 0) + pm(i, j + 1, 0));
        const double cff = 0.25 * (d2(i, j + 1, d2k) + d2(i, j, d2k)) * pnom_v;
        FE1 = cff * (Hz(i, j + 1, k) + hz0) * (rhs(i, j + 1, k) - r0); FE1 *= mskv(i, j + 1, 0);
    }
    const double cff = dt * pm(i, j, 0) * pn(i, j, 0);
    st(i, j, k) += cff * ((FX1 - FX0) + (FE1 - FE0));
}
extern "C" int rh_t3dmix(rh_ctx *c, rh_box bx, rh_array4 state, rh_array4 state_rhs, rh_array4 diff2, rh_array4 Hz, rh_array4 pm,
                         rh_array4 pn, rh_array4 msku, rh_array4 mskv, double dt, int ncomp)
{
    TScope t(c, "t3dmix");
    const rh_box g1 = bx_grow(bx, 1, 1, 0), g12 = bx_slab(g1, 2, 0);
    if (ncomp > state.ncomp || ncomp > state_rhs.ncomp) return rh_fail("rh_t3dmix: ncomp exceeds the arrays");
    RH_NEED(state, bx); RH_NEED(state_rhs, g1); RH_NEED(Hz, g1); RH_NEED(pm, g12); RH_NEED(pn, g12);
    { rh_box m = bx_slab(bx, 2, 0); m.hi[0] += 1; RH_NEED(msku, m); }
    { rh_box m = bx_slab(bx, 2, 0); m.hi[1] += 1; RH_NEED(mskv, m); }
    for (int n = 0; n < ncomp; n++) {
        rh_box dk = g12; dk.lo[2] = dk.hi[2] = n; RH_NEED(diff2, dk);
        k_t3dmix<<<grid3(bx), rh_block(), 0, c->stream>>>(mkB(bx), mkA(state, n), mkA(state_rhs, n), mkA(diff2), n, mkA(Hz), mkA(pm), mkA(pn),
            mkA(msku), mkA(mskv), dt);
        RH_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::coriolis + REMORA::rhs_uv_3d, REMORA_coriolis.cpp:10-40 and REMORA_rhs_uv_3d.cpp:31-385.
// One upward sweep per column: Coriolis term, horizontal advection (third-order upstream, Gadv = -1/4, or centred second
// order), fourth-order vertical advection with W averaged to the velocity point, and the vertical integrals rufrc / rvfrc
// with surface and bottom stress.  UFx/UFe/VFx/VFe/FC never touch memory.
struct UVGeom { int dlox, dhix, dloy, dhiy, perx, pery; };

template <int C2, class TA, class TB>
__device__ __forceinline__ double f_UFx(const TA &u, const TB &Hu, int i, int j, int k, const UVGeom &g)
{
    if (C2) return 0.25 * (u(i, j, k) + u(i + 1, j, k)) * (Hu(i, j, k) + Hu(i + 1, j, k));
    const double Gadv = -0.25;
    const double um = u(i - 1, j, k), u0 = u(i, j, k), up = u(i + 1, j, k), upp = u(i + 2, j, k);
    const double hm = Hu(i - 1, j, k), h0 = Hu(i, j, k), hp = Hu(i + 1, j, k), hpp = Hu(i + 2, j, k);
    const double cff1 = u0 + up;
    double uxx_i = um - 2.0 * u0 + up, uxx_ip1 = u0 - 2.0 * up + upp;
    double Huxx_i = hm - 2.0 * h0 + hp, Huxx_ip1 = h0 - 2.0 * hp + hpp;
    if (i == g.dlox && !g.perx) { uxx_i = uxx_ip1; Huxx_i = Huxx_ip1; }
    else if (i == g.dhix && !g.perx) { uxx_ip1 = uxx_i; Huxx_ip1 = Huxx_i; }
    const double cff = (cff1 > 0.0) ? uxx_i : uxx_ip1;
    const double Huon_avg = (h0 + hp);
    return 0.25 * (cff1 + Gadv * cff) * (Huon_avg + 0.5 * Gadv * (Huxx_i + Huxx_ip1));
}
template <int C2, class TA, class TB>
__device__ __forceinline__ double f_UFe(const TA &u, const TB &Hv, int i, int j, int k, const UVGeom &g)
{
    if (C2) return 0.25 * (u(i, j - 1, k) + u(i, j, k)) * (Hv(i - 1, j, k) + Hv(i, j, k));
    const double Gadv = -0.25;
    const double umm = u(i, j - 2, k), um = u(i, j - 1, k), u0 = u(i, j, k), up = u(i, j + 1, k);
    const double hmm = Hv(i - 2, j, k), hm = Hv(i - 1, j, k), h0 = Hv(i, j, k), hp = Hv(i + 1, j, k);
    const double cff1 = u0 + um;
    const double cff2 = h0 + hm;
    double uee_jm1 = umm - 2.0 * um + u0, uee_j = um - 2.0 * u0 + up;
    if (j == g.dloy && !g.pery) uee_jm1 = uee_j;
    else if (j == g.dhiy + 1 && !g.pery) uee_j = uee_jm1;
    const double cff = (cff2 > 0.0) ? uee_jm1 : uee_j;
    const double Hvxx_i = hm - 2.0 * h0 + hp, Hvxx_im1 = hmm - 2.0 * hm + h0;
    return 0.25 * (cff1 + Gadv * cff) * (cff2 + Gadv * 0.5 * (Hvxx_i + Hvxx_im1));
}
template <int C2, class TA, class TB>
__device__ __forceinline__ double f_VFx(const TA &v, const TB &Hu, int i, int j, int k, const UVGeom &g)
{
    if (C2) return 0.25 * (v(i - 1, j, k) + v(i, j, k)) * (Hu(i, j - 1, k) + Hu(i, j, k));
    const double Gadv = -0.25;
    const double vmm = v(i - 2, j, k), vm = v(i - 1, j, k), v0 = v(i, j, k), vp = v(i + 1, j, k);
    const double hmm = Hu(i, j - 2, k), hm = Hu(i, j - 1, k), h0 = Hu(i, j, k), hp = Hu(i, j + 1, k);
    const double cff1 = v0 + vm;
    const double cff2 = h0 + hm;
    double vxx_im1 = vmm - 2.0 * vm + v0, vxx_i = vm - 2.0 * v0 + vp;
    if (i == g.dlox && !g.perx) vxx_im1 = vxx_i;
    else if (i == g.dhix + 1 && !g.perx) vxx_i = vxx_im1;
    const double cff = (cff2 > 0.0) ? vxx_im1 : vxx_i;
    const double Huee_j = hm - 2.0 * h0 + hp, Huee_jm1 = hmm - 2.0 * hm + h0;
    return 0.25 * (cff1 + Gadv * cff) * (cff2 + Gadv * 0.5 * (Huee_j + Huee_jm1));
}
template <int C2, class TA, class TB>
__device__ __forceinline__ double f_VFe(const TA &v, const TB &Hv, int i, int j, int k, const UVGeom &g)
{
    if (C2) return 0.25 * (v(i, j, k) + v(i, j + 1, k)) * (Hv(i, j, k) + Hv(i, j + 1, k));
    const double Gadv = -0.25;
    const double vm = v(i, j - 1, k), v0 = v(i, j, k), vp = v(i, j + 1, k), vpp = v(i, j + 2, k);
    const double hm = Hv(i, j - 1, k), h0 = Hv(i, j, k), hp = Hv(i, j + 1, k), hpp = Hv(i, j + 2, k);
    const double cff1 = v0 + vp;
    double vee_j = vm - 2.0 * v0 + vp, vee_jp1 = v0 - 2.0 * vp + vpp;
    double Hvee_j = hm - 2.0 * h0 + hp, Hvee_jp1 = h0 - 2.0 * hp + hpp;
    if (j == g.dloy && !g.pery) { vee_j = vee_jp1; Hvee_j = Hvee_jp1; }
    else if (j == g.dhiy && !g.pery) { vee_jp1 = vee_j; Hvee_jp1 = Hvee_j; }
    const double cff = (cff1 > 0.0) ? vee_j : vee_jp1;
    return 0.25 * (cff1 + Gadv * cff) * (h0 + hp + 0.5 * Gadv * (Hvee_j + Hvee_jp1));
}
// vertical momentum flux through k-face kk at a u point (di=1,dj=0) or v point (di=0,dj=1), REMORA_rhs_uv_3d.cpp:168-198,319-345
template <int C2>
__device__ __forceinline__ double f_FCw(const A4 &q, const A4 &W, int i, int j, int kk, int N, int di, int dj)
{
    if (kk <= 0 || kk >= N + 1) return 0.0;
    if (C2) return 0.25 * (q(i, j, kk - 1) + q(i, j, kk)) * (W(i, j, kk) + W(i - di, j - dj, kk));
    const double cff1 = 9.0 / 16.0, cff2 = 1.0 / 16.0;
    const double wpart = (cff1 * (W(i, j, kk) + W(i - di, j - dj, kk)) - cff2 * (W(i + di, j + dj, kk) + W(i - 2 * di, j - 2 * dj, kk)));
    if (kk > 1 && kk <= N - 1)
        return (cff1 * (q(i, j, kk - 1) + q(i, j, kk)) - cff2 * (q(i, j, kk - 2) + q(i, j, kk + 1))) * wpart;
    if (kk == N)     // also the k = 1 face when N == 1 is excluded by rh_create (nz >= 3)
        return (cff1 * (q(i, j, N - 1) + q(i, j, N)) - cff2 * (q(i, j, N - 2) + q(i, j, N))) * wpart;
    return (cff1 * (q(i, j, 0) + q(i, j, 1)) - cff2 * (q(i, j, 0) + q(i, j, 2))) * wpart;          // kk == 1
}

template <int C2>
__global__ void k_rhs_uv_3d(B3 xb, B3 yb, B3 un, A4 uold, A4 vold, A4 ru, A4 rv, A4 rufrc, A4 rvfrc, A4 sustr, A4 svstr,
                            A4 bustr, A4 bvstr, A4 Huon, A4 Hvom, A4 pm, A4 pn, A4 W, A4 Hz, A4 fcor, int use_cor, int nrhs, int N, UVGeom g)
{
    RH_IJ(un);
    if (inb2(xb, i, j)) {
        double fm0 = 0.0, fm1 = 0.0;
        if (use_cor) { fm0 = fcor(i, j, 0) * (1.0 / (pm(i, j, 0) * pn(i, j, 0))); fm1 = fcor(i - 1, j, 0) * (1.0 / (pm(i - 1, j, 0) * pn(i - 1, j, 0))); }
        double frc = rufrc(i, j, 0);
        const double om_u = 2.0 / (pm(i - 1, j, 0) + pm(i, j, 0)), on_u = 2.0 / (pn(i - 1, j, 0) + pn(i, j, 0));
        const double cffs = om_u * on_u;
        const double s_top = sustr(i, j, 0) * cffs, s_bot = -bustr(i, j, 0) * cffs;
        double FCk = 0.0;
        for (int k = 0; k <= N; k++) {
            double r = ru(i, j, k, nrhs);
            if (use_cor) {
                const double UFx_i = 0.5 * Hz(i, j, k) * fm0 * (vold(i, j, k) + vold(i, j + 1, k));
                const double UFx_im1 = 0.5 * Hz(i - 1, j, k) * fm1 * (vold(i - 1, j, k) + vold(i - 1, j + 1, k));
                r += 0.5 * (UFx_i + UFx_im1);
            }
            r -= ((f_UFx<C2>(uold, Huon, i, j, k, g) - f_UFx<C2>(uold, Huon, i - 1, j, k, g)) +
                  (f_UFe<C2>(uold, Hvom, i, j + 1, k, g) - f_UFe<C2>(uold, Hvom, i, j, k, g)));
            const double FCk1 = f_FCw<C2>(uold, W, i, j, k + 1, N, 1, 0);
            const double cff = FCk1 - FCk;
            r -= cff;
            ru(i, j, k, nrhs) = r;
            frc += r;
            const double c1 = (k == N) ? s_top : 0.0, c2 = (k == 0) ? s_bot : 0.0;
            frc += c1 + c2;
            FCk = FCk1;
        }
        rufrc(i, j, 0) = frc;
    }
    if (inb2(yb, i, j)) {
        double fm0 = 0.0, fm1 = 0.0;
        if (use_cor) { fm0 = fcor(i, j, 0) * (1.0 / (pm(i, j, 0) * pn(i, j, 0))); fm1 = fcor(i, j - 1, 0) * (1.0 / (pm(i, j - 1, 0) * pn(i, j - 1, 0))); }
        double frc = rvfrc(i, j, 0);
        const double om_v = 2.0 / (pm(i, j - 1, 0) + pm(i, j, 0)), on_v = 2.0 / (pn(i, j - 1, 0) + pn(i, j, 0));
        const double cffs = om_v * on_v;
        const double s_top = svstr(i, j, 0) * cffs, s_bot = -bvstr(i, j, 0) * cffs;
        double FCk = 0.0;
        for (int k = 0; k <= N; k++) {
            double r = rv(i, j, k, nrhs);
            if (use_cor) {
                const double VFe_j = 0.5 * Hz(i, j, k) * fm0 * (uold(i, j, k) + uold(i + 1, j, k));
                const double VFe_jm1 = 0.5 * Hz(i, j - 1, k) * fm1 * (uold(i, j - 1, k) + uold(i + 1, j - 1, k));
                r -= 0.5 * (VFe_j + VFe_jm1);
            }
            r -= ((f_VFx<C2>(vold, Huon, i + 1, j, k, g) - f_VFx<C2>(vold, Huon, i, j, k, g)) +
                  (f_VFe<C2>(vold, Hvom, i, j, k, g) - f_VFe<C2>(vold, Hvom, i, j - 1, k, g)));
            const double FCk1 = f_FCw<C2>(vold, W, i, j, k + 1, N, 0, 1);
            const double cff = FCk1 - FCk;
            r -= cff;
            rv(i, j, k, nrhs) = r;
            frc += r;
            const double c1 = (k == N) ? s_top : 0.0, c2 = (k == 0) ? s_bot : 0.0;
            frc += c1 + c2;
            FCk = FCk1;
        }
        rvfrc(i, j, 0) = frc;
    }
}
// REMORA::coriolis as a leaf of its own, REMORA_coriolis.cpp:10-40 (inside the drivers it is folded into the momentum kernels).
// Also serves the 2-D call of advance_2d.cpp:530 (Hz := Drhs, k = 0 slabs).
__global__ void k_coriolis(B3 xb, B3 yb, B3 un, A4 uold, A4 vold, A4 ru, A4 rv, A4 Hz, A4 fomn, int nrhs, int nr)
{
    RH_IJ(un);
    const int k = un.lo2 + RH_KZ;
    if (inb2(xb, i, j)) {
        const double UFx_i = 0.5 * Hz(i, j, k) * fomn(i, j, 0) * (vold(i, j, k, nrhs) + vold(i, j + 1, k, nrhs));
        const double UFx_im1 = 0.5 * Hz(i - 1, j, k) * fomn(i - 1, j, 0) * (vold(i - 1, j, k, nrhs) + vold(i - 1, j + 1, k, nrhs));
        ru(i, j, k, nr) += 0.5 * (UFx_i + UFx_im1);
    }
    if (inb2(yb, i, j)) {
        const double VFe_j = 0.5 * Hz(i, j, k) * fomn(i, j, 0) * (uold(i, j, k, nrhs) + uold(i + 1, j, k, nrhs));
        const double VFe_jm1 = 0.5 * Hz(i, j - 1, k) * fomn(i, j - 1, 0) * (uold(i, j - 1, k, nrhs) + uold(i + 1, j - 1, k, nrhs));
        rv(i, j, k, nr) -= 0.5 * (VFe_j + VFe_jm1);
    }
}
extern "C" int rh_coriolis(rh_ctx *c, rh_box xbx, rh_box ybx, rh_array4 uold, rh_array4 vold, rh_array4 ru, rh_array4 rv,
                           rh_array4 Hz, rh_array4 fomn, int nrhs, int nr)
{
    TScope t(c, "coriolis");
    if (xbx.lo[2] != ybx.lo[2] || xbx.hi[2] != ybx.hi[2]) return rh_fail("rh_coriolis: xbx and ybx must span the same levels");
    if (nrhs < 0 || nrhs >= uold.ncomp || nrhs >= vold.ncomp || nr < 0 || nr >= ru.ncomp || nr >= rv.ncomp) return rh_fail("rh_coriolis: component out of range");
    { rh_box q = xbx; q.lo[0] -= 1; q.hi[1] += 1; RH_NEED(vold, q); RH_NEED(Hz, bx_growlo_i(xbx)); RH_NEED(fomn, bx_slab(bx_growlo_i(xbx), 2, 0)); }
    { rh_box q = ybx; q.lo[1] -= 1; q.hi[0] += 1; RH_NEED(uold, q); RH_NEED(Hz, bx_growlo_j(ybx)); RH_NEED(fomn, bx_slab(bx_growlo_j(ybx), 2, 0)); }
    RH_NEED(ru, xbx); RH_NEED(rv, ybx);
    const rh_box un = bx_union(xbx, ybx);
    k_coriolis<<<grid3(un), rh_block(), 0, c->stream>>>(mkB(xbx), mkB(ybx), mkB(un), mkA(uold), mkA(vold), mkA(ru), mkA(rv), mkA(Hz), mkA(fomn), nrhs, nr);
    RH_LAUNCH_CHECK(); return 0;
}

// ---- LDS-tiled, software-pipelined form of k_rhs_uv_3d (same arithmetic).  uold, vold, Huon, Hvom, W(k+1) are staged with a
// halo of 2 and Hz with a halo of 1; each thread loads its own point of every field once per level, one level ahead of its use;
// the vertical flux uses a 4-level register window.  k-invariant per-thread values (Coriolis factors, stress terms) sit in LDS.
__device__ __forceinline__ double win_FCw(double qm1, double q0, double qp1, double qp2, double wpart, int kk, int N, int c2, double w0, double w1)
{   // vertical momentum flux through face kk from the window q(kk-2), q(kk-1), q(kk), q(kk+1) = qm1, q0, qp1, qp2  (:168-198, :319-345)
    if (kk <= 0 || kk >= N + 1) return 0.0;
    if (c2) return 0.25 * (q0 + qp1) * (w0 + w1);
    const double cff1 = 9.0 / 16.0, cff2 = 1.0 / 16.0;
    if (kk > 1 && kk <= N - 1) return (cff1 * (q0 + qp1) - cff2 * (qm1 + qp2)) * wpart;
    if (kk == N) return (cff1 * (q0 + qp1) - cff2 * (qm1 + qp1)) * wpart;       // q(N-1)+q(N), q(N-2)+q(N)
    return (cff1 * (q0 + qp1) - cff2 * (q0 + qp2)) * wpart;                      // kk == 1: q(0)+q(1), q(0)+q(2)
}
// SH = 1: the two fluxes through faces normal to x are evaluated ONCE.  UFx of the rho point west of a u point and VFx of the psi point
// of a v point are what a lane forms; the ones east of it are its neighbour lane's (a DPP wave shift: no LDS traffic), so a thread
// evaluates six fluxes instead of eight and reads 14 LDS operands fewer per level.  Lane 63 has no eastern neighbour: tiles advance by
// 63 columns and its point belongs to the next tile (for the 2049 faces of the bench box that is 33 tile columns either way).
// The shifted value is the same function of the same operands the lane would have evaluated itself: bit-identical.
__device__ __forceinline__ double from_east_lane(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false);      // wave_shl:1 -- lane l takes lane l + 1's value
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int C2, int SH>
__global__ void __launch_bounds__(RT_BX * RT_BY, 2)
k_rhs_uv_tiled(B3 xb, B3 yb, B3 un, A4 uold, A4 vold, A4 ru, A4 rv, A4 rufrc, A4 rvfrc, A4 sustr, A4 svstr, A4 bustr, A4 bvstr,
               A4 Huon, A4 Hvom, A4 pm, A4 pn, A4 W, A4 Hz, A4 fcor, int use_cor, int nrhs, int N, UVGeom g)
{
    typedef Tile<2, 2> T2; typedef Tile<1, 1> T1; typedef TView<2, 2> V2; typedef TView<1, 1> V1;
    constexpr int NT = RT_BX * RT_BY, SET = 5 * T2::SIZE + T1::SIZE, NINV = 8;
    __shared__ double lds[2 * SET + NINV * NT];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * RT_BX + tx;
    double *inv = lds + 2 * SET + tid;
    const RhTile rh_t_ = rh_tile_patch<8, 4>();          // one block per CU (109 KB LDS): 32 resident per XCD
    const int i0 = un.lo0 + rh_t_.x * (SH ? RT_BX - 1 : RT_BX), j0 = un.lo1 + rh_t_.y * RT_BY;
    const int i = i0 + tx, j = j0 + ty;
    const bool mine = !SH || tx < RT_BX - 1;
    const bool au = mine && inb2(xb, i, j), av = mine && inb2(yb, i, j);
    // k-invariants: Coriolis factors fomn at (i,j), (i-1,j), (i,j-1); stress terms; running vertical integrals
    {
        double fm0 = 0.0, fmx = 0.0, fmy = 0.0, su_t = 0.0, su_b = 0.0, sv_t = 0.0, sv_b = 0.0;
        if (au || av) { if (use_cor) fm0 = fcor(i, j, 0) * (1.0 / (pm(i, j, 0) * pn(i, j, 0))); }
        if (au) {
            if (use_cor) fmx = fcor(i - 1, j, 0) * (1.0 / (pm(i - 1, j, 0) * pn(i - 1, j, 0)));
            const double om_u = 2.0 / (pm(i - 1, j, 0) + pm(i, j, 0)), on_u = 2.0 / (pn(i - 1, j, 0) + pn(i, j, 0));
            const double cffs = om_u * on_u;
            su_t = sustr(i, j, 0) * cffs; su_b = -bustr(i, j, 0) * cffs;
        }
        if (av) {
            if (use_cor) fmy = fcor(i, j - 1, 0) * (1.0 / (pm(i, j - 1, 0) * pn(i, j - 1, 0)));
            const double om_v = 2.0 / (pm(i, j - 1, 0) + pm(i, j, 0)), on_v = 2.0 / (pn(i, j - 1, 0) + pn(i, j, 0));
            const double cffs = om_v * on_v;
            sv_t = svstr(i, j, 0) * cffs; sv_b = -bvstr(i, j, 0) * cffs;
        }
        inv[0 * NT] = fm0; inv[1 * NT] = fmx; inv[2 * NT] = fmy; inv[3 * NT] = su_t; inv[4 * NT] = su_b; inv[5 * NT] = sv_t; inv[6 * NT] = sv_b;
    }
    double frcu = au ? rufrc(i, j, 0) : 0.0, frcv = av ? rvfrc(i, j, 0) : 0.0;
    const HaloCell hu = halo_cell<2, 2>(uold, i0, j0, tx, ty), hv = halo_cell<2, 2>(vold, i0, j0, tx, ty), hHu = halo_cell<2, 2>(Huon, i0, j0, tx, ty),
                   hHv = halo_cell<2, 2>(Hvom, i0, j0, tx, ty), hW = halo_cell<2, 2>(W, i0, j0, tx, ty), hHz = halo_cell<1, 1>(Hz, i0, j0, tx, ty);
    // column cursors of the own point and of this thread's halo cell (rh_tile.h); every one can be dereferenced (points outside their
    // array read a dummy), so the loads of a level are unconditional
    const ColPtr cu = col_ptr_any(uold, i, j), cv = col_ptr_any(vold, i, j), cHu = col_ptr_any(Huon, i, j), cHv = col_ptr_any(Hvom, i, j), cHz = col_ptr_any(Hz, i, j),
                 cW = col_ptr_any(W, i, j), cru = au ? col_ptr(ru, i, j, nrhs) : ColPtr{ru.p, 0}, crv = av ? col_ptr(rv, i, j, nrhs) : ColPtr{rv.p, 0};
    const ColPtr qu = halo_ptr_any(uold, hu), qv = halo_ptr_any(vold, hv), qHu = halo_ptr_any(Huon, hHu), qHv = halo_ptr_any(Hvom, hHv), qW = halo_ptr_any(W, hW), qHz = halo_ptr_any(Hz, hHz);
    // vertical windows q(k-1), q(k), q(k+1), q(k+2) of the own point (the top entry arrives with the level's operand set)
    double um1 = 0.0, u0 = col_ldu(cu, 0), up1 = col_ldu(cu, 1 <= N ? 1 : N), up2 = 0.0;
    double vm1 = 0.0, v0 = col_ldu(cv, 0), vp1 = col_ldu(cv, 1 <= N ? 1 : N), vp2 = 0.0;
    // The operands of level k, requested TWO levels ahead (two register sets, the loop unrolled by two; see k_tracers_tiled): own
    // points and halo cells of Huon, Hvom, Hz, W(k + 1), the halo cells of uold / vold, ru / rv, and the own point of uold / vold at
    // k + 2, the new top of the vertical window.
    struct Lv { double un, vn, Hu, Hv, hz, W1, uh, vh, Huh, Hvh, Wh, hzh, ru, rv; };
    auto load = [&](int k) -> Lv {
        const int kc = k <= N ? k : N, k2 = k + 2 <= N ? k + 2 : N;
        Lv q;
        q.un = col_ldu(cu, k2); q.vn = col_ldu(cv, k2);
        q.Hu = col_ldu(cHu, kc); q.Hv = col_ldu(cHv, kc); q.hz = col_ldu(cHz, kc); q.W1 = col_ldu(cW, kc + 1);
        q.uh = col_ldu(qu, kc); q.vh = col_ldu(qv, kc); q.Huh = col_ldu(qHu, kc); q.Hvh = col_ldu(qHv, kc);
        q.Wh = col_ldu(qW, kc + 1); q.hzh = col_ldu(qHz, kc);
        q.ru = col_ldu(cru, kc); q.rv = col_ldu(crv, kc);
        return q;
    };
    double FCu = 0.0, FCv = 0.0;
    auto level = [&](int k, Lv &S, int knext) {
        double *set = lds + (k & 1) * SET;
        double *su = set, *sv = su + T2::SIZE, *sHu = sv + T2::SIZE, *sHv = sHu + T2::SIZE, *sW = sHv + T2::SIZE, *sHz = sW + T2::SIZE;
        tile_put<2, 2>(su, tx, ty, u0, hu, S.uh); tile_put<2, 2>(sv, tx, ty, v0, hv, S.vh);
        tile_put<2, 2>(sHu, tx, ty, S.Hu, hHu, S.Huh); tile_put<2, 2>(sHv, tx, ty, S.Hv, hHv, S.Hvh);
        tile_put<2, 2>(sW, tx, ty, S.W1, hW, S.Wh); tile_put<1, 1>(sHz, tx, ty, S.hz, hHz, S.hzh);
        const double hz = S.hz, W1 = S.W1, rru = S.ru, rrv = S.rv;
        up2 = S.un; vp2 = S.vn;
        S = load(knext);                                     // level k + 2 into the set that has just been consumed
        __syncthreads();
        const V2 tu{su, i0, j0}, tv{sv, i0, j0}, tHu{sHu, i0, j0}, tHv{sHv, i0, j0}, tW{sW, i0, j0};
        const V1 tHz{sHz, i0, j0};
        double UFx_w = 0.0, UFx_e = 0.0, VFx_w = 0.0, VFx_e = 0.0;
        if (SH) {
            UFx_w = f_UFx<C2>(tu, tHu, i - 1, j, 0, g); VFx_w = f_VFx<C2>(tv, tHu, i, j, 0, g);
            UFx_e = from_east_lane(UFx_w); VFx_e = from_east_lane(VFx_w);
        }
        if (au) {
            double r = rru;
            if (use_cor) {
                const double UFx_i = 0.5 * hz * inv[0 * NT] * (v0 + tv(i, j + 1, 0));
                const double UFx_im1 = 0.5 * tHz(i - 1, j, 0) * inv[1 * NT] * (tv(i - 1, j, 0) + tv(i - 1, j + 1, 0));
                r += 0.5 * (UFx_i + UFx_im1);
            }
            if (!SH) { UFx_e = f_UFx<C2>(tu, tHu, i, j, 0, g); UFx_w = f_UFx<C2>(tu, tHu, i - 1, j, 0, g); }
            r -= ((UFx_e - UFx_w) +
                  (f_UFe<C2>(tu, tHv, i, j + 1, 0, g) - f_UFe<C2>(tu, tHv, i, j, 0, g)));
            const double w0 = W1, w1 = tW(i - 1, j, 0);
            const double wpart = C2 ? 0.0 : ((9.0 / 16.0) * (w0 + w1) - (1.0 / 16.0) * (tW(i + 1, j, 0) + tW(i - 2, j, 0)));
            const double FCk1 = win_FCw(um1, u0, up1, up2, wpart, k + 1, N, C2, w0, w1);
            const double cff = FCk1 - FCu;
            r -= cff;
            col_st(cru, k, r);
            frcu += r;
            const double c1 = (k == N) ? inv[3 * NT] : 0.0, c2 = (k == 0) ? inv[4 * NT] : 0.0;
            frcu += c1 + c2;
            FCu = FCk1;
        }
        if (av) {
            double r = rrv;
            if (use_cor) {
                const double VFe_j = 0.5 * hz * inv[0 * NT] * (u0 + tu(i + 1, j, 0));
                const double VFe_jm1 = 0.5 * tHz(i, j - 1, 0) * inv[2 * NT] * (tu(i, j - 1, 0) + tu(i + 1, j - 1, 0));
                r -= 0.5 * (VFe_j + VFe_jm1);
            }
            if (!SH) { VFx_e = f_VFx<C2>(tv, tHu, i + 1, j, 0, g); VFx_w = f_VFx<C2>(tv, tHu, i, j, 0, g); }
            r -= ((VFx_e - VFx_w) +
                  (f_VFe<C2>(tv, tHv, i, j, 0, g) - f_VFe<C2>(tv, tHv, i, j - 1, 0, g)));
            const double w0 = W1, w1 = tW(i, j - 1, 0);
            const double wpart = C2 ? 0.0 : ((9.0 / 16.0) * (w0 + w1) - (1.0 / 16.0) * (tW(i, j + 1, 0) + tW(i, j - 2, 0)));
            const double FCk1 = win_FCw(vm1, v0, vp1, vp2, wpart, k + 1, N, C2, w0, w1);
            const double cff = FCk1 - FCv;
            r -= cff;
            col_st(crv, k, r);
            frcv += r;
            const double c1 = (k == N) ? inv[5 * NT] : 0.0, c2 = (k == 0) ? inv[6 * NT] : 0.0;
            frcv += c1 + c2;
            FCv = FCk1;
        }
        um1 = u0; u0 = up1; up1 = up2; vm1 = v0; v0 = vp1; vp1 = vp2;
    };
    // (no test inside the loop: a level that may be skipped would make every wait count after it conservative)
    Lv A = load(0), B = load(1);
    int k = 0;
    for (; k + 1 <= N; k += 2) { level(k, A, k + 2); level(k + 1, B, k + 3); }
    if (k <= N) level(k, A, k + 2);
    if (au) rufrc(i, j, 0) = frcu;
    if (av) rvfrc(i, j, 0) = frcv;
}

extern "C" int rh_rhs_uv_3d(rh_ctx *c, rh_box xbx, rh_box ybx, rh_array4 uold, rh_array4 vold, rh_array4 ru, rh_array4 rv,
                            rh_array4 rufrc, rh_array4 rvfrc, rh_array4 sustr, rh_array4 svstr, rh_array4 bustr, rh_array4 bvstr,
                            rh_array4 Huon, rh_array4 Hvom, rh_array4 pm, rh_array4 pn, rh_array4 W, rh_array4 Hz, rh_array4 fcor,
                            int use_coriolis, int nrhs, int N)
{
    TScope t(c, "rhs_uv_3d");
    xbx.lo[2] = 0; xbx.hi[2] = N; ybx.lo[2] = 0; ybx.hi[2] = N;
    if (nrhs != 0) return rh_fail("rh_rhs_uv_3d: nrhs must be 0 (uold/vold are single-component, REMORA_setup_step.cpp:43)");
    // stencil extents: u: i-2..i+2 / j-2..j+2 ; Huon i-2..i+2 ; Hvom i-2..i+1, j..j+1 ; W i-2..i+1
    { rh_box q = bx_grow(xbx, 2, 2, 0); RH_NEED(uold, q); }
    { rh_box q = bx_grow(ybx, 2, 2, 0); RH_NEED(vold, q); }
    { rh_box q = xbx; q.lo[0] -= 2; q.hi[0] += 2; RH_NEED(Huon, q); q = xbx; q.lo[0] -= 2; q.hi[0] += 1; q.hi[1] += 1; RH_NEED(Hvom, q); }
    { rh_box q = ybx; q.lo[1] -= 2; q.hi[1] += 2; RH_NEED(Hvom, q); q = ybx; q.lo[1] -= 2; q.hi[1] += 1; q.hi[0] += 1; RH_NEED(Huon, q); }
    { rh_box q = xbx; q.lo[0] -= 2; q.hi[0] += 1; q.hi[2] = N + 1; RH_NEED(W, q); q = ybx; q.lo[1] -= 2; q.hi[1] += 1; q.hi[2] = N + 1; RH_NEED(W, q); }
    { rh_box q = xbx; q.lo[0] -= 1; q.hi[1] += 1; RH_NEED(vold, q); RH_NEED(Hz, bx_growlo_i(xbx)); q = ybx; q.lo[1] -= 1; q.hi[0] += 1; RH_NEED(uold, q); RH_NEED(Hz, bx_growlo_j(ybx)); }
    const rh_box x2 = bx_slab(xbx, 2, 0), y2 = bx_slab(ybx, 2, 0);
    RH_NEED(ru, xbx); RH_NEED(rv, ybx); RH_NEED(rufrc, x2); RH_NEED(rvfrc, y2); RH_NEED(sustr, x2); RH_NEED(svstr, y2); RH_NEED(bustr, x2); RH_NEED(bvstr, y2);
    RH_NEED(pm, bx_growlo_i(x2)); RH_NEED(pn, bx_growlo_i(x2)); RH_NEED(pm, bx_growlo_j(y2)); RH_NEED(pn, bx_growlo_j(y2));
    if (use_coriolis) { RH_NEED(fcor, bx_growlo_i(x2)); RH_NEED(fcor, bx_growlo_j(y2)); }
    UVGeom g = {c->dom.lo[0], c->dom.hi[0], c->dom.lo[1], c->dom.hi[1], c->prm.periodic[0], c->prm.periodic[1]};
    const rh_box un = bx_union(xbx, ybx);
#define RH_UVARGS mkB(xbx), mkB(ybx), mkB(un), mkA(uold), mkA(vold), mkA(ru), mkA(rv), \
        mkA(rufrc), mkA(rvfrc), mkA(sustr), mkA(svstr), mkA(bustr), mkA(bvstr), mkA(Huon), mkA(Hvom), mkA(pm), mkA(pn), mkA(W), mkA(Hz), mkA(fcor), \
        use_coriolis, nrhs, N, g
    if (c->unfused) {           // RH_UNFUSED=1: the direct-from-global form, kept as the A/B reference of the tiled kernel
        if (c->prm.uv_c2) k_rhs_uv_3d<1><<<grid2(un), rh_block(), 0, c->stream>>>(RH_UVARGS); else k_rhs_uv_3d<0><<<grid2(un), rh_block(), 0, c->stream>>>(RH_UVARGS);
    } else {
        dim3 gs = rt_grid(un); gs.x = (unsigned)((un.hi[0] - un.lo[0] + RT_BX - 1) / (RT_BX - 1));        // tiles advance by 63 columns (SH = 1)
        if (c->no_xshare) { if (c->prm.uv_c2) k_rhs_uv_tiled<1, 0><<<rt_grid(un), rt_block(), 0, c->stream>>>(RH_UVARGS); else k_rhs_uv_tiled<0, 0><<<rt_grid(un), rt_block(), 0, c->stream>>>(RH_UVARGS); }
        else if (c->prm.uv_c2) k_rhs_uv_tiled<1, 1><<<gs, rt_block(), 0, c->stream>>>(RH_UVARGS); else k_rhs_uv_tiled<0, 1><<<gs, rt_block(), 0, c->stream>>>(RH_UVARGS);
    }
#undef RH_UVARGS
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::uv3dmix, REMORA_uv3dmix.cpp:6-149: harmonic viscosity along s-surfaces.  Stress at rho points (UFx, VFe) and at
// psi points (UFe, VFx); shared with the 2-D call of the barotropic step (REMORA_advance_2d.cpp:536-538) in rh_step2d.hip.
#include "rh_uvmix.h"

__global__ void k_uv3dmix(B3 xb, B3 yb, B3 un, A4 u, A4 v, A4 uold, A4 vold, A4 rufrc, A4 rvfrc, A4 visc2_p, A4 visc2_r, A4 Hz,
                          A4 pm, A4 pn, A4 mskp, int N, double dt)
{
    RH_IJ(un);
    if (inb2(xb, i, j)) {
        const MixR r0 = mix_r_metrics(pm, pn, visc2_r, i, j), r1 = mix_r_metrics(pm, pn, visc2_r, i - 1, j);
        const MixP p0 = mix_p_metrics(pm, pn, visc2_p, mskp, i, j), p1 = mix_p_metrics(pm, pn, visc2_p, mskp, i, j + 1);
        const double spm = pm(i - 1, j, 0) + pm(i, j, 0), spn = pn(i - 1, j, 0) + pn(i, j, 0);
        const double cff = dt * 0.25 * spm * spn;
        double frc = rufrc(i, j, 0);
        for (int k = 0; k <= N; k++) {
            const double UFx0 = r0.on_r * r0.on_r * r0.visc * mix_r_cff(r0, Hz, uold, vold, i, j, k);
            const double UFx1 = r1.on_r * r1.on_r * r1.visc * mix_r_cff(r1, Hz, uold, vold, i - 1, j, k);
            const double UFe0 = p0.om_p * p0.om_p * p0.visc * mix_p_cff(p0, Hz, uold, vold, i, j, k);
            const double UFe1 = p1.om_p * p1.om_p * p1.visc * mix_p_cff(p1, Hz, uold, vold, i, j + 1, k);
            const double cff1 = 0.5 * spn * (UFx0 - UFx1);
            const double cff2 = 0.5 * spm * (UFe1 - UFe0);
            const double cff3 = cff * (cff1 + cff2);
            u(i, j, k) = u(i, j, k) + cff3;
            frc += cff1 + cff2;
        }
        rufrc(i, j, 0) = frc;
    }
    if (inb2(yb, i, j)) {
        const MixR r0 = mix_r_metrics(pm, pn, visc2_r, i, j), r1 = mix_r_metrics(pm, pn, visc2_r, i, j - 1);
        const MixP p0 = mix_p_metrics(pm, pn, visc2_p, mskp, i, j), p1 = mix_p_metrics(pm, pn, visc2_p, mskp, i + 1, j);
        const double spm = pm(i, j, 0) + pm(i, j - 1, 0), spn = pn(i, j, 0) + pn(i, j - 1, 0);
        const double cff = dt * 0.25 * spm * spn;
        const double spn2 = pn(i, j - 1, 0) + pn(i, j, 0), spm2 = pm(i, j - 1, 0) + pm(i, j, 0);
        double frc = rvfrc(i, j, 0);
        for (int k = 0; k <= N; k++) {
            const double VFe0 = r0.om_r * r0.om_r * r0.visc * mix_r_cff(r0, Hz, uold, vold, i, j, k);
            const double VFe1 = r1.om_r * r1.om_r * r1.visc * mix_r_cff(r1, Hz, uold, vold, i, j - 1, k);
            const double VFx0 = p0.on_p * p0.on_p * p0.visc * mix_p_cff(p0, Hz, uold, vold, i, j, k);
            const double VFx1 = p1.on_p * p1.on_p * p1.visc * mix_p_cff(p1, Hz, uold, vold, i + 1, j, k);
            const double cff1 = 0.5 * spn2 * (VFx1 - VFx0);
            const double cff2 = 0.5 * spm2 * (VFe0 - VFe1);
            const double cff3 = cff * (cff1 - cff2);
            v(i, j, k) = v(i, j, k) + cff3;
            frc += cff1 - cff2;
        }
        rvfrc(i, j, 0) = frc;
    }
}
// LDS-tiled, k-marching form of k_uv3dmix (same arithmetic).  The direct form evaluates every stress twice and asks the fabric
// for 15 doubles per cell where 5 are distinct (profiles/r1_v3_traffic.json).  Here a block stages uold, vold, Hz of one level
// (64 x 16 footprint, halo 1); each thread evaluates the stresses of ITS rho point and ITS psi point once and parks them in LDS;
// the u and v points combine their own values with the neighbour's.  Blocks overlap by one column and one row (stride 63 x 15;
// as in k2d_uvbar), so no thread computes a halo stress.  Input and stress tiles are double-buffered on the parity of k and the
// update runs one level behind the stresses: one barrier per level.
#define M3_SX (RT_BX - 1)
#define M3_BY 16          // rows per block: the staged footprint is (M3_BY + 2) / (M3_BY - 1) rows per row of output (8: 1.43, 16: 1.2)
#define M3_SY (M3_BY - 1)
__global__ void __launch_bounds__(RT_BX * M3_BY)
k_uv3dmix_tiled(B3 xb, B3 yb, B3 un, A4 u, A4 v, A4 uold, A4 vold, A4 rufrc, A4 rvfrc, A4 visc2_p, A4 visc2_r, A4 Hz,
                A4 pm, A4 pn, A4 mskp, int N, double dt)
{
    typedef Tile<1, 1, M3_BY> T1; typedef TView<1, 1, M3_BY> V1;
    constexpr int NT = RT_BX * M3_BY, SET = 3 * T1::SIZE, FSET = 4 * NT;
    __shared__ double lds[2 * SET + 2 * FSET];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * RT_BX + tx;
    const RhTile rh_t_ = rh_tile();
    const int i0 = un.lo0 - 1 + rh_t_.x * M3_SX, j0 = un.lo1 - 1 + rh_t_.y * M3_SY;
    const int i = i0 + tx, j = j0 + ty;
    // rho points un.lo-1 .. un.hi and psi points un.lo .. un.hi+1 carry a stress some velocity point needs
    const bool ar = i >= un.lo0 - 1 && i <= un.hi0 && j >= un.lo1 - 1 && j <= un.hi1;
    const bool ap = i >= un.lo0 && i <= un.hi0 + 1 && j >= un.lo1 && j <= un.hi1 + 1;
    const bool au = tx >= 1 && ty <= M3_BY - 2 && inb2(xb, i, j), av = tx <= RT_BX - 2 && ty >= 1 && inb2(yb, i, j);
    MixR r; MixP p; memset(&r, 0, sizeof r); memset(&p, 0, sizeof p);
    double fr_x = 0.0, fr_e = 0.0, fp_e = 0.0, fp_x = 0.0;
    if (ar) { r = mix_r_metrics(pm, pn, visc2_r, i, j); fr_x = r.on_r * r.on_r * r.visc; fr_e = r.om_r * r.om_r * r.visc; }
    if (ap) { p = mix_p_metrics(pm, pn, visc2_p, mskp, i, j); fp_e = p.om_p * p.om_p * p.visc; fp_x = p.on_p * p.on_p * p.visc; }
    double cu = 0.0, hspn_u = 0.0, hspm_u = 0.0, frcu = 0.0, cv = 0.0, hspn_v = 0.0, hspm_v = 0.0, frcv = 0.0;
    if (au) {
        const double spm = pm(i - 1, j, 0) + pm(i, j, 0), spn = pn(i - 1, j, 0) + pn(i, j, 0);
        cu = dt * 0.25 * spm * spn; hspn_u = 0.5 * spn; hspm_u = 0.5 * spm; frcu = rufrc(i, j, 0);
    }
    if (av) {
        const double spm = pm(i, j, 0) + pm(i, j - 1, 0), spn = pn(i, j, 0) + pn(i, j - 1, 0);
        const double spn2 = pn(i, j - 1, 0) + pn(i, j, 0), spm2 = pm(i, j - 1, 0) + pm(i, j, 0);
        cv = dt * 0.25 * spm * spn; hspn_v = 0.5 * spn2; hspm_v = 0.5 * spm2; frcv = rvfrc(i, j, 0);
    }
    const HaloCell hu = halo_cell<1, 1, M3_BY>(uold, i0, j0, tx, ty), hv = halo_cell<1, 1, M3_BY>(vold, i0, j0, tx, ty), hh = halo_cell<1, 1, M3_BY>(Hz, i0, j0, tx, ty);
    double u0 = ld_in(uold, i, j, 0), v0 = ld_in(vold, i, j, 0), hz = ld_in(Hz, i, j, 0);
    double uh = halo_ld(uold, hu, 0), vh = halo_ld(vold, hv, 0), hzh = halo_ld(Hz, hh, 0);
    double un_k = au ? u(i, j, 0) : 0.0, vn_k = av ? v(i, j, 0) : 0.0;        // the fields being updated, level of the pending update
    for (int k = 0; k <= N + 1; k++) {
        double *set = lds + (k & 1) * SET, *fl = lds + 2 * SET + (k & 1) * FSET + tid;
        const double *flp = lds + 2 * SET + ((k + 1) & 1) * FSET + tid;       // stresses of level k-1
        double u0_n = 0.0, v0_n = 0.0, hz_n = 0.0, uh_n = 0.0, vh_n = 0.0, hzh_n = 0.0, un_n = 0.0, vn_n = 0.0;
        if (k <= N) {
            tile_put<1, 1>(set, tx, ty, u0, hu, uh); tile_put<1, 1>(set + T1::SIZE, tx, ty, v0, hv, vh); tile_put<1, 1>(set + 2 * T1::SIZE, tx, ty, hz, hh, hzh);
            if (k + 1 <= N) {
                u0_n = ld_in(uold, i, j, k + 1); v0_n = ld_in(vold, i, j, k + 1); hz_n = ld_in(Hz, i, j, k + 1);
                uh_n = halo_ld(uold, hu, k + 1); vh_n = halo_ld(vold, hv, k + 1); hzh_n = halo_ld(Hz, hh, k + 1);
            }
            if (k >= 1) { un_n = au ? u(i, j, k) : 0.0; vn_n = av ? v(i, j, k) : 0.0; }
        }
        __syncthreads();
        if (k <= N) {
            const V1 tu{set, i0, j0}, tv{set + T1::SIZE, i0, j0}, th{set + 2 * T1::SIZE, i0, j0};
            double cr = 0.0, cp = 0.0;
            if (ar) cr = mix_r_cff(r, th, tu, tv, i, j, 0);
            if (ap) cp = mix_p_cff(p, th, tu, tv, i, j, 0);
            fl[0 * NT] = fr_x * cr; fl[1 * NT] = fr_e * cr; fl[2 * NT] = fp_e * cp; fl[3 * NT] = fp_x * cp;     // UFx, VFe, UFe, VFx
        }
        if (k >= 1) {
            const int km = k - 1;
            if (au) {
                const double UFx0 = flp[0 * NT], UFx1 = flp[0 * NT - 1], UFe0 = flp[2 * NT], UFe1 = flp[2 * NT + RT_BX];
                const double cff1 = hspn_u * (UFx0 - UFx1);
                const double cff2 = hspm_u * (UFe1 - UFe0);
                const double cff3 = cu * (cff1 + cff2);
                u(i, j, km) = un_k + cff3;
                frcu += cff1 + cff2;
            }
            if (av) {
                const double VFe0 = flp[1 * NT], VFe1 = flp[1 * NT - RT_BX], VFx0 = flp[3 * NT], VFx1 = flp[3 * NT + 1];
                const double cff1 = hspn_v * (VFx1 - VFx0);
                const double cff2 = hspm_v * (VFe0 - VFe1);
                const double cff3 = cv * (cff1 - cff2);
                v(i, j, km) = vn_k + cff3;
                frcv += cff1 - cff2;
            }
        }
        if (k >= 1) { un_k = un_n; vn_k = vn_n; }
        u0 = u0_n; v0 = v0_n; hz = hz_n; uh = uh_n; vh = vh_n; hzh = hzh_n;
    }
    if (au) rufrc(i, j, 0) = frcu;
    if (av) rvfrc(i, j, 0) = frcv;
}

extern "C" int rh_uv3dmix(rh_ctx *c, rh_box xbx, rh_box ybx, rh_array4 u, rh_array4 v, rh_array4 uold, rh_array4 vold,
                          rh_array4 rufrc, rh_array4 rvfrc, rh_array4 visc2_p, rh_array4 visc2_r, rh_array4 Hz, rh_array4 pm,
                          rh_array4 pn, rh_array4 mskp, int nrhs, int nnew, double dt)
{
    TScope t(c, "uv3dmix");
    const int N = xbx.hi[2] - ybx.lo[2];                                       // REMORA_uv3dmix.cpp:36
    if (nrhs != 0 || nnew != 0) return rh_fail("rh_uv3dmix: nrhs and nnew must be 0 for the 3-D call");
    if (xbx.lo[2] != 0 || ybx.lo[2] != 0) return rh_fail("rh_uv3dmix: boxes must start at k = 0");
    const rh_box un = bx_union(xbx, ybx);
    { rh_box q = bx_grow(un, 1, 1, 0); RH_NEED(uold, q); RH_NEED(vold, q); RH_NEED(Hz, q);
      rh_box q2 = bx_slab(q, 2, 0); RH_NEED(visc2_r, q2); RH_NEED(visc2_p, q2); RH_NEED(mskp, q2);
      q2 = bx_slab(bx_grow(un, 2, 2, 0), 2, 0); RH_NEED(pm, q2); RH_NEED(pn, q2); }
    RH_NEED(u, xbx); RH_NEED(v, ybx); RH_NEED(rufrc, bx_slab(xbx, 2, 0)); RH_NEED(rvfrc, bx_slab(ybx, 2, 0));
    if (c->unfused)
        k_uv3dmix<<<grid2(un), rh_block(), 0, c->stream>>>(mkB(xbx), mkB(ybx), mkB(un), mkA(u), mkA(v), mkA(uold), mkA(vold), mkA(rufrc), mkA(rvfrc),
            mkA(visc2_p), mkA(visc2_r), mkA(Hz), mkA(pm), mkA(pn), mkA(mskp), N, dt);
    else {
        const dim3 grid((un.hi[0] - un.lo[0] + 1 + M3_SX) / M3_SX, (un.hi[1] - un.lo[1] + 1 + M3_SY) / M3_SY, 1);
        k_uv3dmix_tiled<<<grid, dim3(RT_BX, M3_BY, 1), 0, c->stream>>>(mkB(xbx), mkB(ybx), mkB(un), mkA(u), mkA(v), mkA(uold), mkA(vold), mkA(rufrc), mkA(rvfrc),
            mkA(visc2_p), mkA(visc2_r), mkA(Hz), mkA(pm), mkA(pn), mkA(mskp), N, dt);
    }
    RH_LAUNCH_CHECK(); return 0;
}
