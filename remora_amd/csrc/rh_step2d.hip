// rh_step2d.hip -- the barotropic fast step: REMORA::advance_2d (= ROMS step2d), REMORA_advance_2d.cpp:41-757, with
// REMORA::rhs_uv_2d (REMORA_rhs_uv_2d.cpp:20-238) and the 2-D calls of coriolis / uv3dmix folded in.
//
// The reference issues ~35 ParallelFor passes over 13 scratch FABs per call; here a call is three kernels
//   k2d_flux   : Drhs, DUon, DVom                                             (:149-166)
//   k2d_zeta   : time-average accumulation + free-surface step + rzeta          (:293-464)
//   k2d_uvbar  : pressure gradient, advection, Coriolis, viscosity, 3D->2D forcing coupling, ubar/vbar step   (:479-730)
// separated only where a ghost exchange (:170-171) or a neighbour dependency forces it.  One thread per (i,j).
#include "rh_common.h"
#include "rh_uvmix.h"
#include "rh_tile.h"

__global__ void k2d_flux(B3 b3, B3 xg, B3 yg, A4 zeta, A4 h, A4 ubar, A4 vbar, A4 pm, A4 pn, A4 DUon, A4 DVom)
{
    RH_IJ(b3); if (!inb2(b3, i, j)) return;
    const double d0 = zeta(i, j, 0) + h(i, j, 0);          // Drhs, :149-152 (rebuilt by its consumers, never stored)
    if (inb2(xg, i, j)) {
        const double on_u = 2.0 / (pn(i, j, 0) + pn(i - 1, j, 0));
        const double cff1 = 0.5 * on_u * (d0 + (zeta(i - 1, j, 0) + h(i - 1, j, 0)));
        DUon(i, j, 0) = ubar(i, j, 0) * cff1;
    }
    if (inb2(yg, i, j)) {
        const double om_v = 2.0 / (pm(i, j, 0) + pm(i, j - 1, 0));
        const double cff1 = 0.5 * om_v * (d0 + (zeta(i, j - 1, 0) + h(i, j - 1, 0)));
        DVom(i, j, 0) = vbar(i, j, 0) * cff1;
    }
}

struct ZetaPar {
    int avg_mode;            // 0: predictor & first, 1: predictor & not first, 2: corrector
    double wt_a, wt_b;       // avg_mode 0: wt_b = cff2 ; 1: wt_a = cff1_wt1, wt_b = cff2_wt1 ; 2: wt_b = cff2_wt2
    int do_step;             // my_iif < nfast
    int zmode;               // 0: my_iif == 0 (forward Euler), 1: predictor (leap-frog), 2: corrector (Adams-Moulton)
    int predictor;
    int nomask;              // rh_params.masks_all_one: mskr == 1 everywhere, x * 1.0 == x, so the mask is not read
    double dtfast;
};

__global__ void k2d_zeta(B3 g3, B3 xg, B3 yg, B3 t1, ZetaPar zp, A4 zeta_rhs, A4 zeta_stp, A4 zeta_new_o, A4 rzeta_stp, A4 rzeta_pts,
                         A4 rzeta_rhs, A4 h, A4 pm, A4 pn, A4 mskr, A4 rhoS, A4 rhoA, A4 DUon, A4 DVom, A4 Zt_avg1,
                         A4 DU_avg1, A4 DU_avg2, A4 DV_avg1, A4 DV_avg2, A4 zwrk_o)
{
    RH_IJ(g3); if (!inb2(g3, i, j)) return;
    // ---- time averages, :293-356
    if (zp.avg_mode == 0) {
        Zt_avg1(i, j, 0) = 0.0;
        if (inb2(xg, i, j)) { DU_avg1(i, j, 0) = 0.0; DU_avg2(i, j, 0) = zp.wt_b * DUon(i, j, 0); }
        if (inb2(yg, i, j)) { DV_avg1(i, j, 0) = 0.0; DV_avg2(i, j, 0) = zp.wt_b * DVom(i, j, 0); }
    } else if (zp.avg_mode == 1) {
        Zt_avg1(i, j, 0) += zp.wt_a * zeta_rhs(i, j, 0);
        if (inb2(xg, i, j)) { const double d = DUon(i, j, 0); DU_avg1(i, j, 0) += zp.wt_a * d; DU_avg2(i, j, 0) += zp.wt_b * d; }
        if (inb2(yg, i, j)) { const double d = DVom(i, j, 0); DV_avg1(i, j, 0) += zp.wt_a * d; DV_avg2(i, j, 0) += zp.wt_b * d; }
    } else {
        if (inb2(xg, i, j)) DU_avg2(i, j, 0) = DU_avg2(i, j, 0) + zp.wt_b * DUon(i, j, 0);
        if (inb2(yg, i, j)) DV_avg2(i, j, 0) = DV_avg2(i, j, 0) + zp.wt_b * DVom(i, j, 0);
    }
    if (!zp.do_step || !inb2(t1, i, j)) return;
    // ---- free surface, :379-464
    const double div = (DUon(i, j, 0) - DUon(i + 1, j, 0)) + (DVom(i, j, 0) - DVom(i, j + 1, 0));
    const double zs = zeta_stp(i, j, 0);
    const double mr = zp.nomask ? 1.0 : mskr(i, j, 0);
    double zn, zw;
    if (zp.zmode == 0) {
        const double cff1 = zp.dtfast;
        zn = (zs + pm(i, j, 0) * pn(i, j, 0) * cff1 * div) * mr;
        zw = 0.5 * (zs + zn);
    } else if (zp.zmode == 1) {
        const double cff1 = 2.0 * zp.dtfast, cff4 = 4.0 / 25.0, cff5 = 1.0 - 2.0 * cff4;
        zn = (zs + pm(i, j, 0) * pn(i, j, 0) * cff1 * div) * mr;
        zw = cff5 * zeta_rhs(i, j, 0) + cff4 * (zs + zn);
    } else {
        const double cff1 = zp.dtfast * 5.0 / 12.0, cff2 = zp.dtfast * 8.0 / 12.0, cff3 = zp.dtfast * 1.0 / 12.0, cff4 = 2.0 / 5.0, cff5 = 1.0 - cff4;
        const double cff = cff1 * div;
        zn = zs + pm(i, j, 0) * pn(i, j, 0) * (cff + cff2 * rzeta_stp(i, j, 0) - cff3 * rzeta_pts(i, j, 0));
        zn *= mr;
        zw = cff5 * zn + cff4 * zeta_rhs(i, j, 0);
    }
    zwrk_o(i, j, 0) = zw;        // Dnew, gzeta, gzeta2, gzetaSA (:392-394 ...) are rebuilt from zwrk / rhoS / rhoA / zeta(knew) / h in k2d_uvbar
    zeta_new_o(i, j, 0) = zn;                                                  // :450-454
    if (zp.predictor) rzeta_rhs(i, j, 0) = div;                                // :459-464
}

struct UVPar {
    int dlox, dhix, dloy, dhiy, perx, pery, uv_c2, use_cor;
    int couple;              // 0: rhs += rufrc ; 1: iic == ntfirst ; 2: iic == ntfirst+1 ; 3: later (first && predictor), :543-624
    int am3;                 // 0: my_iif == 0 || predictor (:643-684) ; 1: corrector (:686-713)
    int predictor;
    int nomask;              // rh_params.masks_all_one: msku, mskv, mskp are 1 everywhere and are not read
    double g, c1, c2, c3;    // step coefficients
};

// 2-D momentum advection fluxes, REMORA_rhs_uv_2d.cpp:80-129,171-216 (u, v = ubar/vbar(krhs))
template <class TA, class TB>
__device__ __forceinline__ double s2_UFx(const TA &u, const TB &DU, int i, int j, const UVPar &g)
{
    if (g.uv_c2) return 0.25 * (DU(i, j, 0) + DU(i + 1, j, 0)) * (u(i, j, 0) + u(i + 1, j, 0));
    const double um = u(i - 1, j, 0), u0 = u(i, j, 0), up = u(i + 1, j, 0), upp = u(i + 2, j, 0);
    const double hm = DU(i - 1, j, 0), h0 = DU(i, j, 0), hp = DU(i + 1, j, 0), hpp = DU(i + 2, j, 0);
    double uxx_i = um - 2.0 * u0 + up, uxx_ip1 = u0 - 2.0 * up + upp;
    double Huxx_i = hm - 2.0 * h0 + hp, Huxx_ip1 = h0 - 2.0 * hp + hpp;
    if (i == g.dlox && !g.perx) { uxx_i = uxx_ip1; Huxx_i = Huxx_ip1; }
    else if (i == g.dhix && !g.perx) { uxx_ip1 = uxx_i; Huxx_ip1 = Huxx_i; }
    const double cff = 1.0 / 6.0;
    const double ubar_avg = u0 + up;
    return 0.25 * (ubar_avg - cff * (uxx_i + uxx_ip1)) * (h0 + hp - cff * (Huxx_i + Huxx_ip1));
}
template <class TA, class TB>
__device__ __forceinline__ double s2_UFe(const TA &u, const TB &DV, int i, int j, const UVPar &g)
{
    if (g.uv_c2) return 0.25 * (DV(i, j, 0) + DV(i - 1, j, 0)) * (u(i, j, 0) + u(i, j - 1, 0));
    const double umm = u(i, j - 2, 0), um = u(i, j - 1, 0), u0 = u(i, j, 0), up = u(i, j + 1, 0);
    const double hmm = DV(i - 2, j, 0), hm = DV(i - 1, j, 0), h0 = DV(i, j, 0), hp = DV(i + 1, j, 0);
    double uee_j = um - 2.0 * u0 + up, uee_jm1 = umm - 2.0 * um + u0;
    const double Hvxx_i = hm - 2.0 * h0 + hp, Hvxx_im1 = hmm - 2.0 * hm + h0;
    if (j == g.dloy && !g.pery) uee_jm1 = uee_j;
    else if (j == g.dhiy + 1 && !g.pery) uee_j = uee_jm1;
    const double cff = 1.0 / 6.0;
    const double cff1 = u0 + um;
    const double cff2 = h0 + hm;
    return 0.25 * (cff1 - (uee_j + uee_jm1) * cff) * (cff2 - cff * (Hvxx_i + Hvxx_im1));
}
template <class TA, class TB>
__device__ __forceinline__ double s2_VFx(const TA &v, const TB &DU, int i, int j, const UVPar &g)
{
    if (g.uv_c2) return 0.25 * (DU(i, j, 0) + DU(i, j - 1, 0)) * (v(i, j, 0) + v(i - 1, j, 0));
    const double cff = 1.0 / 6.0;
    const double vmm = v(i - 2, j, 0), vm = v(i - 1, j, 0), v0 = v(i, j, 0), vp = v(i + 1, j, 0);
    const double hmm = DU(i, j - 2, 0), hm = DU(i, j - 1, 0), h0 = DU(i, j, 0), hp = DU(i, j + 1, 0);
    double vxx_i = vm - 2.0 * v0 + vp, vxx_im1 = vmm - 2.0 * vm + v0;
    const double Huee_j = hm - 2.0 * h0 + hp, Huee_jm1 = hmm - 2.0 * hm + h0;
    if (i == g.dlox && !g.perx) vxx_im1 = vxx_i;
    else if (i == g.dhix + 1 && !g.perx) vxx_i = vxx_im1;
    const double cff1 = v0 + vm;
    const double cff2 = h0 + hm;
    return 0.25 * (cff1 - (vxx_i + vxx_im1) * cff) * (cff2 - cff * (Huee_j + Huee_jm1));
}
template <class TA, class TB>
__device__ __forceinline__ double s2_VFe(const TA &v, const TB &DV, int i, int j, const UVPar &g)
{
    if (g.uv_c2) return 0.25 * (DV(i, j, 0) + DV(i, j + 1, 0)) * (v(i, j, 0) + v(i, j + 1, 0));
    const double vm = v(i, j - 1, 0), v0 = v(i, j, 0), vp = v(i, j + 1, 0), vpp = v(i, j + 2, 0);
    const double hm = DV(i, j - 1, 0), h0 = DV(i, j, 0), hp = DV(i, j + 1, 0), hpp = DV(i, j + 2, 0);
    double vee_j = vm - 2.0 * v0 + vp, vee_jp1 = v0 - 2.0 * vp + vpp;
    double Hvee_j = hm - 2.0 * h0 + hp, Hvee_jp1 = h0 - 2.0 * hp + hpp;
    if (j == g.dloy && !g.pery) { vee_j = vee_jp1; Hvee_j = Hvee_jp1; }
    else if (j == g.dhiy && !g.pery) { vee_jp1 = vee_j; Hvee_jp1 = Hvee_j; }
    const double cff = 1.0 / 6.0;
    const double cff1 = v0 + vp;
    return 0.25 * (cff1 - (vee_j + vee_jp1) * cff) * (h0 + hp - cff * (Hvee_j + Hvee_jp1));
}

// Drhs = zeta(krhs) + h, rebuilt where it is used (bitwise the value of the reference's Drhs scratch FAB, :149-152)
// mask view that skips the load when the masks are known to be 1 (x * 1.0 == x bit for bit)
struct MaskView { A4 m; int one; __device__ __forceinline__ double operator()(int i, int j, int) const { return one ? 1.0 : m(i, j, 0); } };
struct DrhsView { A4 z, h; __device__ __forceinline__ double operator()(int i, int j, int) const { return z(i, j, 0) + h(i, j, 0); } };

// Flux sharing through LDS.  Every flux of the reference's scratch FABs (UFx, UFe, VFx, VFe of rhs_uv_2d and of uv3dmix, the
// Coriolis terms, gzeta / gzeta2 / gzetaSA) belongs to a rho point or a psi point and is needed by two velocity points.  Phase 1:
// each thread evaluates the 18 shared quantities of ITS point once (metric terms and their divisions included) and parks them in
// LDS; phase 2: the u and v points combine their own values with the neighbour's (i-1 / j-1 for rho, i+1 / j+1 for psi).
// Blocks overlap by one column and one row (stride 63 x 7 for 64 x 8 threads), so no halo pass is needed: u is produced for
// tx >= 1, ty <= 6 and v for tx <= 62, ty >= 1.  gzeta, gzeta2, gzetaSA, Dnew, Dstp, Drhs never travel through memory.
#define U_BY 8
#define U_SX (RT_BX - 1)
#define U_SY (U_BY - 1)
enum { Q_UFX, Q_VFE, Q_MUFX, Q_MVFE, Q_CORU, Q_CORV, Q_GZ, Q_GZ2, Q_GSA, Q_H, Q_ZW, Q_RA, Q_PM, Q_PN, Q_UFE, Q_VFX, Q_MUFE, Q_MVFX, Q_N };
__global__ void __launch_bounds__(RT_BX * U_BY)
k2d_uvbar(B3 xb, B3 yb, B3 un, B3 rb, UVPar P, A4 ub_rhs, A4 vb_rhs, A4 ub_stp, A4 vb_stp, A4 ub_new, A4 vb_new,
          A4 rubar_rhs, A4 rvbar_rhs, A4 rubar_stp, A4 rvbar_stp, A4 rubar_pts, A4 rvbar_pts,
          A4 zeta_rhs, A4 zeta_stp, A4 zeta_new, A4 h, A4 pm, A4 pn, A4 fcor, A4 rhoS, A4 rhoA, A4 msku, A4 mskv, A4 mskp, A4 visc2_p, A4 visc2_r,
          A4 rufrc, A4 rvfrc, A4 ru2d, A4 rv2d, A4 DUon, A4 DVom, A4 zwrk)
{
    constexpr int NT = RT_BX * U_BY;
    __shared__ double lds[Q_N * NT];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * RT_BX + tx;
    const RhTile rh_t_ = rh_tile();
    const int i = un.lo0 - 1 + rh_t_.x * U_SX + tx, j = un.lo1 - 1 + rh_t_.y * U_SY + ty;
    double *q = lds + tid;                                  // q[Q * NT]: this thread's slot of quantity Q
    const DrhsView Drhs{zeta_rhs, h};
    const double fac = 1000.0 / 1025.0;
    double h0 = 0, zw0 = 0, rA0 = 0, pm0 = 0, pn0 = 0, gz0 = 0, gz20 = 0, gSA0 = 0;
    double UFx = 0, VFe = 0, mUFx = 0, mVFe = 0, corU = 0, corV = 0, UFe = 0, VFx = 0, mUFe = 0, mVFx = 0;
    if (inb2(rb, i, j)) {                                   // rho point (i,j), cells lo-1 .. hi+1
        h0 = h(i, j, 0); zw0 = zwrk(i, j, 0); rA0 = rhoA(i, j, 0); pm0 = pm(i, j, 0); pn0 = pn(i, j, 0);
        const double rS0 = rhoS(i, j, 0);
        gz0 = (fac + rS0) * zw0; gz20 = gz0 * zw0; gSA0 = zw0 * (rS0 - rA0);               // :392-394 and the like
        UFx = s2_UFx(ub_rhs, DUon, i, j, P);                                                 // rhs_uv_2d :80-102
        VFe = s2_VFe(vb_rhs, DVom, i, j, P);                                                 // :193-216
        const MixR r = mix_r_metrics(pm, pn, visc2_r, i, j);                                 // uv3dmix :48-59, :103-115 with Hz := Drhs
        const double cffr = mix_r_cff(r, Drhs, ub_rhs, vb_rhs, i, j, 0);
        mUFx = r.on_r * r.on_r * r.visc * cffr;
        mVFe = r.om_r * r.om_r * r.visc * cffr;
        if (P.use_cor) {                                                                     // coriolis :19-27,30-38 with Hz := Drhs
            const double fm = fcor(i, j, 0) * (1.0 / (pm0 * pn0));
            const double d = Drhs(i, j, 0);
            corU = 0.5 * d * fm * (vb_rhs(i, j, 0) + vb_rhs(i, j + 1, 0));
            corV = 0.5 * d * fm * (ub_rhs(i, j, 0) + ub_rhs(i + 1, j, 0));
        }
    }
    if (inb2(un, i, j)) {                                   // psi point (i,j), nodes lo .. hi+1
        UFe = s2_UFe(ub_rhs, DVom, i, j, P);                                                 // :107-129
        VFx = s2_VFx(vb_rhs, DUon, i, j, P);                                                 // :171-191
        const MixP p = mix_p_metrics(pm, pn, visc2_p, MaskView{mskp, P.nomask}, i, j);       // uv3dmix :61-78, :117-134
        const double cffp = mix_p_cff(p, Drhs, ub_rhs, vb_rhs, i, j, 0);
        mUFe = p.om_p * p.om_p * p.visc * cffp;
        mVFx = p.on_p * p.on_p * p.visc * cffp;
    }
    q[Q_UFX * NT] = UFx; q[Q_VFE * NT] = VFe; q[Q_MUFX * NT] = mUFx; q[Q_MVFE * NT] = mVFe; q[Q_CORU * NT] = corU; q[Q_CORV * NT] = corV;
    q[Q_GZ * NT] = gz0; q[Q_GZ2 * NT] = gz20; q[Q_GSA * NT] = gSA0; q[Q_H * NT] = h0; q[Q_ZW * NT] = zw0; q[Q_RA * NT] = rA0;
    q[Q_PM * NT] = pm0; q[Q_PN * NT] = pn0; q[Q_UFE * NT] = UFe; q[Q_VFX * NT] = VFx; q[Q_MUFE * NT] = mUFe; q[Q_MVFX * NT] = mVFx;
    __syncthreads();
    const double cffg = 0.5 * P.g, third = 1.0 / 3.0;
    if (tx >= 1 && ty <= U_BY - 2 && inb2(xb, i, j)) {
        const double *w = q - 1, *n = q + RT_BX;            // west rho point (i-1,j), north psi point (i,j+1)
        const double hx = w[Q_H * NT], zwx = w[Q_ZW * NT], rAx = w[Q_RA * NT], pmx = w[Q_PM * NT], pnx = w[Q_PN * NT];
        const double on_u = 2.0 / (pn0 + pnx);
        double rhs = cffg * on_u *                                                           // :479-492
            ((hx + h0) * (w[Q_GZ * NT] - gz0) +
             (hx - h0) * (w[Q_GSA * NT] + gSA0 + third * (rAx - rA0) * (zwx - zw0)) +
             (w[Q_GZ2 * NT] - gz20));
        {
            const double cff1 = UFx - w[Q_UFX * NT];
            const double cff2 = n[Q_UFE * NT] - UFe;
            rhs -= (cff1 + cff2);
        }
        if (P.use_cor) rhs += 0.5 * (corU + w[Q_CORU * NT]);
        {
            const double cff1 = 0.5 * (pnx + pn0) * (mUFx - w[Q_MUFX * NT]);
            const double cff2 = 0.5 * (pmx + pm0) * (n[Q_MUFE * NT] - mUFe);
            rhs += cff1 + cff2;
        }
        if (P.couple == 0) rhs += rufrc(i, j, 0);                                            // :543-624
        else {
            double f = rufrc(i, j, 0);
            if (P.couple == 1) { f -= rhs; rhs += f; ru2d(i, j, 0, 0) = f; }
            else if (P.couple == 2) {
                f = f - rhs;
                const double r0 = ru2d(i, j, 0, 0);
                rhs = rhs + 1.5 * f - 0.5 * r0;
                ru2d(i, j, 0, 1) = r0; ru2d(i, j, 0, 0) = f;
            } else {
                const double c1 = 23.0 / 12.0, c2 = 16.0 / 12.0, c3 = 5.0 / 12.0;
                f = f - rhs;
                const double r0 = ru2d(i, j, 0, 0), r1 = ru2d(i, j, 0, 1);
                rhs = rhs + c1 * f - c2 * r0 + c3 * r1;
                ru2d(i, j, 0, 1) = r0; ru2d(i, j, 0, 0) = f;
            }
            rufrc(i, j, 0) = f;
        }
        const double Dstp0 = zeta_stp(i, j, 0) + h0, Dstp1 = zeta_stp(i - 1, j, 0) + hx;     // :626-713
        const double cff = (pm0 + pmx) * (pn0 + pnx);
        const double Dnew_avg = 1.0 / ((zeta_new(i, j, 0) + h0) + (zeta_new(i - 1, j, 0) + hx));
        const double mu = P.nomask ? 1.0 : msku(i, j, 0);
        if (!P.am3) ub_new(i, j, 0) = (ub_stp(i, j, 0) * (Dstp0 + Dstp1) + cff * P.c1 * rhs) * Dnew_avg * mu;
        else ub_new(i, j, 0) = (ub_stp(i, j, 0) * (Dstp0 + Dstp1) +
                                cff * (P.c1 * rhs + P.c2 * rubar_stp(i, j, 0) - P.c3 * rubar_pts(i, j, 0))) * Dnew_avg * mu;
        if (P.predictor) rubar_rhs(i, j, 0) = rhs;                                           // :721-730
    }
    if (tx <= RT_BX - 2 && ty >= 1 && inb2(yb, i, j)) {
        const double *sp = q - RT_BX, *e = q + 1;           // south rho point (i,j-1), east psi point (i+1,j)
        const double hy = sp[Q_H * NT], zwy = sp[Q_ZW * NT], rAy = sp[Q_RA * NT], pmy = sp[Q_PM * NT], pny = sp[Q_PN * NT];
        const double om_v = 2.0 / (pm0 + pmy);
        double rhs = cffg * om_v *
            ((hy + h0) * (sp[Q_GZ * NT] - gz0) +
             (hy - h0) * (sp[Q_GSA * NT] + gSA0 + third * (rAy - rA0) * (zwy - zw0)) +
             (sp[Q_GZ2 * NT] - gz20));
        {
            const double cff1 = e[Q_VFX * NT] - VFx;
            const double cff2 = VFe - sp[Q_VFE * NT];
            rhs -= (cff1 + cff2);
        }
        if (P.use_cor) rhs -= 0.5 * (corV + sp[Q_CORV * NT]);
        {
            const double cff1 = 0.5 * (pny + pn0) * (e[Q_MVFX * NT] - mVFx);
            const double cff2 = 0.5 * (pmy + pm0) * (mVFe - sp[Q_MVFE * NT]);
            rhs += cff1 - cff2;
        }
        if (P.couple == 0) rhs += rvfrc(i, j, 0);
        else {
            double f = rvfrc(i, j, 0);
            if (P.couple == 1) { f -= rhs; rhs += f; rv2d(i, j, 0, 0) = f; }
            else if (P.couple == 2) {
                f = f - rhs;
                const double r0 = rv2d(i, j, 0, 0);
                rhs = rhs + 1.5 * f - 0.5 * r0;
                rv2d(i, j, 0, 1) = r0; rv2d(i, j, 0, 0) = f;
            } else {
                const double c1 = 23.0 / 12.0, c2 = 16.0 / 12.0, c3 = 5.0 / 12.0;
                f = f - rhs;
                const double r0 = rv2d(i, j, 0, 0), r1 = rv2d(i, j, 0, 1);
                rhs = rhs + c1 * f - c2 * r0 + c3 * r1;
                rv2d(i, j, 0, 1) = r0; rv2d(i, j, 0, 0) = f;
            }
            rvfrc(i, j, 0) = f;
        }
        const double Dstp0 = zeta_stp(i, j, 0) + h0, Dstp1 = zeta_stp(i, j - 1, 0) + hy;
        const double cff = (pm0 + pmy) * (pn0 + pny);
        const double Dnew_avg = 1.0 / ((zeta_new(i, j, 0) + h0) + (zeta_new(i, j - 1, 0) + hy));
        const double mv = P.nomask ? 1.0 : mskv(i, j, 0);
        if (!P.am3) vb_new(i, j, 0) = (vb_stp(i, j, 0) * (Dstp0 + Dstp1) + cff * P.c1 * rhs) * Dnew_avg * mv;
        else vb_new(i, j, 0) = (vb_stp(i, j, 0) * (Dstp0 + Dstp1) +
                                cff * (P.c1 * rhs + P.c2 * rvbar_stp(i, j, 0) - P.c3 * rvbar_pts(i, j, 0))) * Dnew_avg * mv;
        if (P.predictor) rvbar_rhs(i, j, 0) = rhs;
    }
}

// ---- the whole call in one kernel -------------------------------------------------------------------------------
// k2d_flux + k2d_zeta + k2d_uvbar of one advance_2d call fused: DUon, DVom, Drhs, zwrk, Dnew, gzeta* never reach memory (the
// three-kernel form moves ~67 doubles per column and call, this one ~35).  Each block first rebuilds what it needs of
// Drhs / DUon / DVom on its 64 x 8 points plus a halo of 2 (68 x 12 tiles in LDS, same expressions as k2d_flux, so the values are
// bitwise those of the scratch FABs), steps the free surface on its own points, and then runs the two phases of k2d_uvbar with
// the tiles in place of the arrays.  The tiles alias the flux slots (they are dead once phase 1 has its values in registers).
// Points of tbxp1 are owned by exactly one thread (blocks overlap by one row / column): only the owner accumulates the time
// averages and stores zeta(knew) / rzeta.  The ghost frame outside tbxp1 gets its averages from k2d_frame.
// The FillBoundary of DUon / DVom (:170-171) is NOT redundant: a ghost face rebuilt from ghost ubar / vbar can differ from its
// owner's value (the BC routines skip the outermost ghost row, REMORA_BoundaryConditions_xvel.cpp:56-134, so e.g. a wall face there
// keeps its pre-BC value).  Where that FillBoundary has work to do (a periodic direction, or neighbour ranks) the ghost faces are
// therefore built and filled the reference's way first -- k2d_flux_frame on the rim + halo_fill_boundary -- and the tiles take their
// ghost faces from memory (ghost_from_mem); the valid faces are always rebuilt here.
#define F_PX (RT_BX + 4)
#define F_PY (U_BY + 4)
#define F_N  (F_PX * F_PY)
struct FTile { const double *s; int i0, j0; __device__ __forceinline__ double operator()(int i, int j, int) const { return s[(j - j0) * F_PX + (i - i0)]; } };
enum { Q_ZN = Q_N, QF_N };
enum { T_D, T_U, T_V, T_UB, T_VB, T_PM, T_PN, T_N };      // tiles: Drhs, DUon, DVom (rebuilt) ; ubar/vbar(krhs), pm, pn (staged)
// One memory round trip per block: every global load of the kernel -- the tile points (two per thread) and this thread's own
// operands of all later phases -- is issued up front; plain loads stay in flight across the barriers, so the LDS phases run
// while the late operands are still arriving.  All stencil neighbours (ubar, vbar, pm, pn, Drhs, DUon, DVom) come from LDS.
// The kernel touches ~40 arrays; as amrex-style views (base + 3-D box + strides each) their descriptors alone need ~480 scalar
// registers, and the compiler spilled them to vector lanes: 38 % of the vector instructions of the first version were
// v_readlane / v_writelane of spilled scalars.  The arrays come in four shapes only -- rho points with 3 or with 2 ghost cells, u
// faces, v faces -- so the kernel takes ONE shape per group plus bare pointers; equal shapes also share their index arithmetic.
struct S2 { int lo0, lo1, hi0, hi1; unsigned sy; };
struct V2 {
    const S2 &s; double *p;
    __device__ __forceinline__ double &operator()(int i, int j, int) const { return p[(unsigned)(i - s.lo0) + (unsigned)(j - s.lo1) * s.sy]; }
};
__device__ __forceinline__ double ld_in(const V2 &a, int i, int j, int)
{ return (i >= a.s.lo0 && i <= a.s.hi0 && j >= a.s.lo1 && j <= a.s.hi1) ? a(i, j, 0) : 0.0; }
enum { R3_ZRHS, R3_ZSTP, R3_ZNEW, R3_H, R3_FCOR, R3_MSKR, R3_ZT, R3_N };                       // rho points, 3 ghost cells
enum { R2_RZSTP, R2_RZPTS, R2_RZRHS, R2_RHOS, R2_RHOA, R2_VR, R2_VP, R2_N };                   // rho points, 2 ghost cells
enum { FA_RHS, FA_STP, FA_NEW, FA_RRHS, FA_RSTP, FA_RPTS, FA_FRC, FA_R2D0, FA_R2D1, FA_AVG1, FA_AVG2, FA_DON, FA_N };   // per face direction
struct K2Arrays { S2 r3, r2, u2, v2, spm, spn, smu, smv, smp; double *R3[R3_N], *R2[R2_N], *U[FA_N], *V[FA_N], *pm, *pn, *msku, *mskv, *mskp; };
struct B2v { int lo0, lo1, hi0, hi1; };      // the rank's valid cell box; every box of the kernel is this one grown / made nodal
// Even so the arguments (~170 dwords) exceed the 102 scalar registers, and what the load phase fetched but the store phases need --
// output pointers, shapes, the time-stepping weights -- was parked in vector lanes across the whole kernel.  The kernel therefore
// takes ONE struct and the later phases read their scalars again from the kernel-argument segment (scalar loads, served by the
// scalar cache) through a pointer the compiler cannot trace back to the argument, so nothing has to stay live from phase to phase.
struct K2Args { B2v vb; UVPar P; ZetaPar zp; K2Arrays A; int ghost_from_mem; int ntile_x, ntile_y, nframe;
                double bc_ext[4]; int wallbc;
                int part; };       // 0: every tile; 1: interior tiles only; 2: rim tiles only (see k2_tile_is_interior)   // wallbc: all four sides are walls of a one-rank domain: the blocks on the rim also write the ghost cells of the new level (bc_ext: ubar x-lo / x-hi, vbar y-lo / y-hi ext_dir values)    // the launch's first nframe blocks (in tile order) carry the ghost frame, the next ntile_x * ntile_y are tiles
__device__ __forceinline__ const K2Args &k2_args_again()
{
    typedef const K2Args __attribute__((address_space(4))) *KP;
    KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();     // the struct is the only argument: offset 0 of the segment
    asm volatile("" : "+s"(kp));
    return *(const K2Args *)kp;
}
// One point of the ghost frame (tbxp3 outside tbxp1), enumerated as in k2d_frame: accumulation of the time averages, :293-356.
__device__ __forceinline__ void k2_frame_point(const K2Args &L, int g)
{
    const B2v &vb = L.vb; const ZetaPar &zp = L.zp; const K2Arrays &A = L.A;
    const B3 g3{vb.lo0 - 3, vb.lo1 - 3, 0, vb.hi0 + 3, vb.hi1 + 3, 0}, t1{vb.lo0 - 1, vb.lo1 - 1, 0, vb.hi0 + 1, vb.hi1 + 1, 0};
    const B3 xg{vb.lo0 - 2, vb.lo1 - 2, 0, vb.hi0 + 3, vb.hi1 + 2, 0}, yg{vb.lo0 - 2, vb.lo1 - 2, 0, vb.hi0 + 2, vb.hi1 + 3, 0};
    const int NX = g3.hi0 - g3.lo0 + 1, NY = g3.hi1 - g3.lo1 + 1, vx = t1.hi0 - t1.lo0 + 1, vy = t1.hi1 - t1.lo1 + 1;
    const int nglo_x = t1.lo0 - g3.lo0, nglo_y = t1.lo1 - g3.lo1;
    const int nrow_part = (NY - vy) * NX, total = nrow_part + vy * (NX - vx);
    if (g >= total) return;
    int i, j;
    if (g < nrow_part) { const int row = g / NX; i = g3.lo0 + g % NX; j = row < nglo_y ? g3.lo1 + row : t1.hi1 + 1 + (row - nglo_y); }
    else { const int r = g - nrow_part, w = NX - vx, row = r / w, cc = r % w; j = t1.lo1 + row; i = cc < nglo_x ? g3.lo0 + cc : t1.hi0 + 1 + (cc - nglo_x); }
    const V2 zeta_rhs{A.r3, A.R3[R3_ZRHS]}, h{A.r3, A.R3[R3_H]}, Zt_avg1{A.r3, A.R3[R3_ZT]}, pm{A.spm, A.pm}, pn{A.spn, A.pn};
    const V2 ub_rhs{A.u2, A.U[FA_RHS]}, DU_avg1{A.u2, A.U[FA_AVG1]}, DU_avg2{A.u2, A.U[FA_AVG2]}, DUon_g{A.u2, A.U[FA_DON]};
    const V2 vb_rhs{A.v2, A.V[FA_RHS]}, DV_avg1{A.v2, A.V[FA_AVG1]}, DV_avg2{A.v2, A.V[FA_AVG2]}, DVom_g{A.v2, A.V[FA_DON]};
    const double zr = zeta_rhs(i, j, 0);
    const double d0 = zr + h(i, j, 0);
    const bool inx = inb2(xg, i, j), iny = inb2(yg, i, j);
    double du = 0.0, dv = 0.0;
    if (inx) {
        if (L.ghost_from_mem) du = DUon_g(i, j, 0);
        else {
            const double on_u = 2.0 / (pn(i, j, 0) + pn(i - 1, j, 0));
            const double cff1 = 0.5 * on_u * (d0 + (zeta_rhs(i - 1, j, 0) + h(i - 1, j, 0)));
            du = ub_rhs(i, j, 0) * cff1;
        }
    }
    if (iny) {
        if (L.ghost_from_mem) dv = DVom_g(i, j, 0);
        else {
            const double om_v = 2.0 / (pm(i, j, 0) + pm(i, j - 1, 0));
            const double cff1 = 0.5 * om_v * (d0 + (zeta_rhs(i, j - 1, 0) + h(i, j - 1, 0)));
            dv = vb_rhs(i, j, 0) * cff1;
        }
    }
    if (zp.avg_mode == 0) {
        Zt_avg1(i, j, 0) = 0.0;
        if (inx) { DU_avg1(i, j, 0) = 0.0; DU_avg2(i, j, 0) = zp.wt_b * du; }
        if (iny) { DV_avg1(i, j, 0) = 0.0; DV_avg2(i, j, 0) = zp.wt_b * dv; }
    } else if (zp.avg_mode == 1) {
        Zt_avg1(i, j, 0) += zp.wt_a * zr;
        if (inx) { DU_avg1(i, j, 0) += zp.wt_a * du; DU_avg2(i, j, 0) += zp.wt_b * du; }
        if (iny) { DV_avg1(i, j, 0) += zp.wt_a * dv; DV_avg2(i, j, 0) += zp.wt_b * dv; }
    } else {
        if (inx) DU_avg2(i, j, 0) = DU_avg2(i, j, 0) + zp.wt_b * du;
        if (iny) DV_avg2(i, j, 0) = DV_avg2(i, j, 0) + zp.wt_b * dv;
    }
}
// A tile is interior when everything it stages (its footprint of F_PX x F_PY points from (I0 - 2, J0 - 2)) lies at least three
// cells inside the rank's valid box: it then reads no ghost cell, no shared face and nothing within reach of the ghost exchange,
// and may run while the exchanges of this call and the ghost fill of the previous one are still in flight.
__host__ __device__ __forceinline__ bool k2_tile_is_interior(const B2v &vb, int I0, int J0)
{ return I0 - 2 >= vb.lo0 + 3 && I0 - 2 + F_PX - 1 <= vb.hi0 - 3 && J0 - 2 >= vb.lo1 + 3 && J0 - 2 + F_PY - 1 <= vb.hi1 - 3; }
// MODE: the two calls that make up all but the first fast step of a baroclinic step are compiled with their time-stepping modes as
// constants -- 1: predictor, not the first step (leap-frog free surface, both sets of time averages, no coupling terms, two-level
// momentum); 2: corrector (Adams-Moulton free surface, second set of averages, three-level momentum) -- so their operand sets and
// arithmetic are straight-line code; 0 reads the modes from the arguments (first fast step, first baroclinic steps).
#define K2_AVG    (MODE == 1 ? 1 : MODE == 2 ? 2 : zp.avg_mode)
#define K2_ZMODE  (MODE == 1 ? 1 : MODE == 2 ? 2 : zp.zmode)
#define K2_ZPRED  (MODE == 1 ? 1 : MODE == 2 ? 0 : zp.predictor)
#define K2_COUPLE (MODE != 0 ? 0 : P.couple)
#define K2_AM3    (MODE == 1 ? 0 : MODE == 2 ? 1 : P.am3)
#define K2_PPRED  (MODE == 1 ? 1 : MODE == 2 ? 0 : P.predictor)
// FIX = 1: masks known to be one and Coriolis on (every pinned case but the f = 0 ones) as constants as well
#define K2_ZNOMASK (FIX ? 1 : zp.nomask)
#define K2_PNOMASK (FIX ? 1 : P.nomask)
#define K2_COR     (FIX ? 1 : P.use_cor)
template <int MODE, int FIX>
__global__ void __launch_bounds__(RT_BX * U_BY)
k2d_fused(K2Args a0)
{
    const B2v &vb = a0.vb; const UVPar &P = a0.P; const ZetaPar &zp = a0.zp; const K2Arrays &A = a0.A; const int ghost_from_mem = a0.ghost_from_mem;
    // advance_2d.cpp's boxes: xbxD, ybxD, their union, tbxp1, tbxp3, xgbx2, ygbx2
    const B3 xb{vb.lo0, vb.lo1, 0, vb.hi0 + 1, vb.hi1, 0}, yb{vb.lo0, vb.lo1, 0, vb.hi0, vb.hi1 + 1, 0}, un{vb.lo0, vb.lo1, 0, vb.hi0 + 1, vb.hi1 + 1, 0};
    const B3 rb{vb.lo0 - 1, vb.lo1 - 1, 0, vb.hi0 + 1, vb.hi1 + 1, 0}, g3{vb.lo0 - 3, vb.lo1 - 3, 0, vb.hi0 + 3, vb.hi1 + 3, 0};
    const B3 xg{vb.lo0 - 2, vb.lo1 - 2, 0, vb.hi0 + 3, vb.hi1 + 2, 0}, yg{vb.lo0 - 2, vb.lo1 - 2, 0, vb.hi0 + 2, vb.hi1 + 3, 0};
    const V2 ub_rhs{A.u2, A.U[FA_RHS]}, ub_stp{A.u2, A.U[FA_STP]}, rubar_stp{A.u2, A.U[FA_RSTP]},
             rubar_pts{A.u2, A.U[FA_RPTS]}, rufrc{A.u2, A.U[FA_FRC]}, ru2d0{A.u2, A.U[FA_R2D0]}, ru2d1{A.u2, A.U[FA_R2D1]}, DU_avg1{A.u2, A.U[FA_AVG1]},
             DU_avg2{A.u2, A.U[FA_AVG2]}, DUon_g{A.u2, A.U[FA_DON]};
    const V2 vb_rhs{A.v2, A.V[FA_RHS]}, vb_stp{A.v2, A.V[FA_STP]}, rvbar_stp{A.v2, A.V[FA_RSTP]},
             rvbar_pts{A.v2, A.V[FA_RPTS]}, rvfrc{A.v2, A.V[FA_FRC]}, rv2d0{A.v2, A.V[FA_R2D0]}, rv2d1{A.v2, A.V[FA_R2D1]}, DV_avg1{A.v2, A.V[FA_AVG1]},
             DV_avg2{A.v2, A.V[FA_AVG2]}, DVom_g{A.v2, A.V[FA_DON]};
    const V2 zeta_rhs{A.r3, A.R3[R3_ZRHS]}, zeta_stp{A.r3, A.R3[R3_ZSTP]}, h{A.r3, A.R3[R3_H]}, fcor{A.r3, A.R3[R3_FCOR]},
             mskr{A.r3, A.R3[R3_MSKR]}, Zt_avg1{A.r3, A.R3[R3_ZT]};
    const V2 rzeta_stp{A.r2, A.R2[R2_RZSTP]}, rzeta_pts{A.r2, A.R2[R2_RZPTS]}, rhoS{A.r2, A.R2[R2_RHOS]},
             rhoA{A.r2, A.R2[R2_RHOA]}, visc2_r{A.r2, A.R2[R2_VR]}, visc2_p{A.r2, A.R2[R2_VP]};
    const V2 pm{A.spm, A.pm}, pn{A.spn, A.pn}, msku{A.smu, A.msku}, mskv{A.smv, A.mskv}, mskp{A.smp, A.mskp};

    constexpr int NT = RT_BX * U_BY;
    __shared__ double lds[QF_N * NT];
    static_assert(T_N * F_N <= QF_N * NT, "tiles alias the flux slots");
    constexpr int TR = (F_N + NT - 1) / NT;                  // tile points per thread
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * RT_BX + tx;
    const RhTile rh_t_ = rh_tile();
    // The first blocks of the launch (in the XCD-aware tile order) accumulate the time averages on the ghost frame -- k2d_frame's work,
    // same expressions, one thread per frame point -- so that they run under the tiles instead of as a tail after them.  Frame points
    // lie outside tbxp1, the tiles' owners inside: no two threads of the launch write the same element, and nothing written by this
    // launch is read by the frame blocks.
    const int lin = rh_t_.y * (int)gridDim.x + rh_t_.x - a0.nframe;
    if (lin < 0) { const K2Args &L = k2_args_again(); k2_frame_point(L, (lin + L.nframe) * NT + tid); return; }
    const int tile_x = lin % a0.ntile_x, tile_y = lin / a0.ntile_x;
    if (tile_y >= a0.ntile_y) return;
    const int I0 = un.lo0 - 1 + tile_x * U_SX, J0 = un.lo1 - 1 + tile_y * U_SY;
    if (a0.part && (a0.part == 1) != k2_tile_is_interior(vb, I0, J0)) return;      // uniform over the block
    const int i = I0 + tx, j = J0 + ty;
    const bool in_rb = inb2(rb, i, j), in_un = inb2(un, i, j);
    const bool owner = in_rb && (tx >= 1 || tile_x == 0) && (ty >= 1 || tile_y == 0);
    const bool is_u = tx >= 1 && ty <= U_BY - 2 && inb2(xb, i, j), is_v = tx <= RT_BX - 2 && ty >= 1 && inb2(yb, i, j);

    // ---- (1) all global loads.  Tile points first: t = tid, tid + NT, ...
    double tz[TR], th[TR], tpm[TR], tpn[TR], tub[TR], tvb[TR], tdu[TR], tdv[TR];
#pragma unroll
    for (int r = 0; r < TR; r++) {
        const int t = tid + r * NT;
        const int y = t / F_PX, x = t - y * F_PX, gi = I0 - 2 + x, gj = J0 - 2 + y;
        const bool on = t < F_N;
        const bool ing = on && inb2(g3, gi, gj);
        tz[r] = ing ? zeta_rhs(gi, gj, 0) : 0.0; th[r] = ing ? h(gi, gj, 0) : 0.0;
        tpm[r] = on ? ld_in(pm, gi, gj, 0) : 0.0; tpn[r] = on ? ld_in(pn, gi, gj, 0) : 0.0;
        tub[r] = on ? ld_in(ub_rhs, gi, gj, 0) : 0.0; tvb[r] = on ? ld_in(vb_rhs, gi, gj, 0) : 0.0;
        tdu[r] = 0.0; tdv[r] = 0.0;
        if (ghost_from_mem && on) {                         // ghost faces: what FillBoundary left in memory
            if (x >= 1 && inb2(xg, gi, gj) && !inb2(xb, gi, gj)) tdu[r] = DUon_g(gi, gj, 0);
            if (y >= 1 && inb2(yg, gi, gj) && !inb2(yb, gi, gj)) tdv[r] = DVom_g(gi, gj, 0);
        }
    }
    // this thread's own operands of the later phases
    double h0 = 0, zr0 = 0, zs0 = 0, rS0 = 0, rA0 = 0, fc0 = 0, vr0 = 0, mr0 = 1.0, rzs = 0, rzp = 0, zt0 = 0, dua1 = 0, dua2 = 0, dva1 = 0, dva2 = 0;
    if (in_rb) {
        h0 = h(i, j, 0); zr0 = zeta_rhs(i, j, 0); zs0 = zeta_stp(i, j, 0); rS0 = rhoS(i, j, 0); rA0 = rhoA(i, j, 0); vr0 = visc2_r(i, j, 0);
        if (K2_COR) fc0 = fcor(i, j, 0);
        if (!K2_ZNOMASK) mr0 = mskr(i, j, 0);
        if (K2_ZMODE == 2) { rzs = rzeta_stp(i, j, 0); rzp = rzeta_pts(i, j, 0); }
        if (owner) {
            if (K2_AVG == 1) { zt0 = Zt_avg1(i, j, 0); dua1 = DU_avg1(i, j, 0); dva1 = DV_avg1(i, j, 0); }
            if (K2_AVG >= 1) { dua2 = DU_avg2(i, j, 0); dva2 = DV_avg2(i, j, 0); }
        }
    }
    double vp0 = 0, mp0 = 1.0;
    if (in_un) { vp0 = visc2_p(i, j, 0); if (!K2_PNOMASK) mp0 = mskp(i, j, 0); }
    double u_frc = 0, u_r0 = 0, u_r1 = 0, u_stp = 0, u_rs = 0, u_rp = 0, u_msk = 1.0, u_zsw = 0;
    if (is_u) {
        u_frc = rufrc(i, j, 0); u_stp = ub_stp(i, j, 0); u_zsw = zeta_stp(i - 1, j, 0);
        if (K2_COUPLE >= 2) u_r0 = ru2d0(i, j, 0);
        if (K2_COUPLE == 3) u_r1 = ru2d1(i, j, 0);
        if (K2_AM3) { u_rs = rubar_stp(i, j, 0); u_rp = rubar_pts(i, j, 0); }
        if (!K2_PNOMASK) u_msk = msku(i, j, 0);
    }
    double v_frc = 0, v_r0 = 0, v_r1 = 0, v_stp = 0, v_rs = 0, v_rp = 0, v_msk = 1.0, v_zss = 0;
    if (is_v) {
        v_frc = rvfrc(i, j, 0); v_stp = vb_stp(i, j, 0); v_zss = zeta_stp(i, j - 1, 0);
        if (K2_COUPLE >= 2) v_r0 = rv2d0(i, j, 0);
        if (K2_COUPLE == 3) v_r1 = rv2d1(i, j, 0);
        if (K2_AM3) { v_rs = rvbar_stp(i, j, 0); v_rp = rvbar_pts(i, j, 0); }
        if (!K2_PNOMASK) v_msk = mskv(i, j, 0);
    }

    // ---- (2) staged tiles: Drhs (advance_2d.cpp:149-152), ubar / vbar(krhs), pm, pn
    double *tl = lds;
#pragma unroll
    for (int r = 0; r < TR; r++) {
        const int t = tid + r * NT;
        if (t < F_N) {
            tl[T_D * F_N + t] = tz[r] + th[r];
            tl[T_UB * F_N + t] = tub[r]; tl[T_VB * F_N + t] = tvb[r]; tl[T_PM * F_N + t] = tpm[r]; tl[T_PN * F_N + t] = tpn[r];
        }
    }
    __syncthreads();
    // ---- (3) DUon, DVom on the tile (:153-166)
#pragma unroll
    for (int r = 0; r < TR; r++) {
        const int t = tid + r * NT;
        if (t < F_N) {
            const int y = t / F_PX, x = t - y * F_PX, gi = I0 - 2 + x, gj = J0 - 2 + y;
            double du = tdu[r], dv = tdv[r];
            if (x >= 1 && inb2(xg, gi, gj) && !(ghost_from_mem && !inb2(xb, gi, gj))) {
                const double on_u = 2.0 / (tpn[r] + tl[T_PN * F_N + t - 1]);
                const double cff1 = 0.5 * on_u * (tl[T_D * F_N + t] + tl[T_D * F_N + t - 1]);
                du = tub[r] * cff1;
            }
            if (y >= 1 && inb2(yg, gi, gj) && !(ghost_from_mem && !inb2(yb, gi, gj))) {
                const double om_v = 2.0 / (tpm[r] + tl[T_PM * F_N + t - F_PX]);
                const double cff1 = 0.5 * om_v * (tl[T_D * F_N + t] + tl[T_D * F_N + t - F_PX]);
                dv = tvb[r] * cff1;
            }
            tl[T_U * F_N + t] = du; tl[T_V * F_N + t] = dv;
        }
    }
    __syncthreads();
    const FTile Drhs{tl + T_D * F_N, I0 - 2, J0 - 2}, DUon{tl + T_U * F_N, I0 - 2, J0 - 2}, DVom{tl + T_V * F_N, I0 - 2, J0 - 2};
    const FTile ubr{tl + T_UB * F_N, I0 - 2, J0 - 2}, vbr{tl + T_VB * F_N, I0 - 2, J0 - 2}, tpmv{tl + T_PM * F_N, I0 - 2, J0 - 2}, tpnv{tl + T_PN * F_N, I0 - 2, J0 - 2};
    // ---- (4) time averages (:293-356) and free surface (:379-464) on this thread's rho point
    double zn0 = 0.0, zw0 = 0.0, pm0 = 0.0, pn0 = 0.0;
    double gz0 = 0, gz20 = 0, gSA0 = 0;
    double UFx = 0, VFe = 0, mUFx = 0, mVFe = 0, corU = 0, corV = 0, UFe = 0, VFx = 0, mUFe = 0, mVFx = 0;
    {   // phases (4), (5): parameters, output pointers and shapes read again from the argument segment (these names shadow the early ones)
    const K2Args &L = k2_args_again();
    const ZetaPar &zp = L.zp; const UVPar &P = L.P;
    const V2 Zt_avg1{L.A.r3, L.A.R3[R3_ZT]}, zeta_new{L.A.r3, L.A.R3[R3_ZNEW]}, rzeta_rhs{L.A.r2, L.A.R2[R2_RZRHS]};
    const V2 DU_avg1{L.A.u2, L.A.U[FA_AVG1]}, DU_avg2{L.A.u2, L.A.U[FA_AVG2]}, DV_avg1{L.A.v2, L.A.V[FA_AVG1]}, DV_avg2{L.A.v2, L.A.V[FA_AVG2]};
    if (in_rb) {
        pm0 = tpmv(i, j, 0); pn0 = tpnv(i, j, 0);
        const double du0 = DUon(i, j, 0), dv0 = DVom(i, j, 0);
        if (owner) {
            if (K2_AVG == 0) {
                Zt_avg1(i, j, 0) = 0.0;
                DU_avg1(i, j, 0) = 0.0; DU_avg2(i, j, 0) = zp.wt_b * du0;
                DV_avg1(i, j, 0) = 0.0; DV_avg2(i, j, 0) = zp.wt_b * dv0;
            } else if (K2_AVG == 1) {
                Zt_avg1(i, j, 0) = zt0 + zp.wt_a * zr0;
                DU_avg1(i, j, 0) = dua1 + zp.wt_a * du0; DU_avg2(i, j, 0) = dua2 + zp.wt_b * du0;
                DV_avg1(i, j, 0) = dva1 + zp.wt_a * dv0; DV_avg2(i, j, 0) = dva2 + zp.wt_b * dv0;
            } else {
                DU_avg2(i, j, 0) = dua2 + zp.wt_b * du0;
                DV_avg2(i, j, 0) = dva2 + zp.wt_b * dv0;
            }
        }
        const double div = (du0 - DUon(i + 1, j, 0)) + (dv0 - DVom(i, j + 1, 0));
        const double zs = zs0, mr = mr0;
        if (K2_ZMODE == 0) {
            const double cff1 = zp.dtfast;
            zn0 = (zs + pm0 * pn0 * cff1 * div) * mr;
            zw0 = 0.5 * (zs + zn0);
        } else if (K2_ZMODE == 1) {
            const double cff1 = 2.0 * zp.dtfast, cff4 = 4.0 / 25.0, cff5 = 1.0 - 2.0 * cff4;
            zn0 = (zs + pm0 * pn0 * cff1 * div) * mr;
            zw0 = cff5 * zr0 + cff4 * (zs + zn0);
        } else {
            const double cff1 = zp.dtfast * 5.0 / 12.0, cff2 = zp.dtfast * 8.0 / 12.0, cff3 = zp.dtfast * 1.0 / 12.0, cff4 = 2.0 / 5.0, cff5 = 1.0 - cff4;
            const double cff = cff1 * div;
            zn0 = zs + pm0 * pn0 * (cff + cff2 * rzs - cff3 * rzp);
            zn0 *= mr;
            zw0 = cff5 * zn0 + cff4 * zr0;
        }
        if (owner) {
            if (!L.wallbc) zeta_new(i, j, 0) = zn0;                            // :450-454
            else if (i >= P.dlox && i <= P.dhix && j >= P.dloy && j <= P.dhiy) {
                // walls: the ghost cells of zeta(knew) end as copies of the nearest valid cell -- what FillPatch's foextrap passes
                // (sides, then the rows over the full width, then 0.5 * (x + x) in the corners; REMORA_BoundaryConditions_cons.cpp:
                // 170-337) leave there -- so the cell next to a wall writes them and the ring of tbxp1 is not stored at all
                zeta_new(i, j, 0) = zn0;
                const int ci = (i == P.dlox) ? -1 : ((i == P.dhix) ? 1 : 0), cj = (j == P.dloy) ? -1 : ((j == P.dhiy) ? 1 : 0);
                if (ci) for (int g = 1; g <= 3; g++) zeta_new(i + ci * g, j, 0) = zn0;
                if (cj) for (int g = 1; g <= 3; g++) zeta_new(i, j + cj * g, 0) = zn0;
                if (ci && cj) for (int gj = 1; gj <= 3; gj++) for (int gi = 1; gi <= 3; gi++) zeta_new(i + ci * gi, j + cj * gj, 0) = zn0;
            }
            if (K2_ZPRED) rzeta_rhs(i, j, 0) = div;                        // :459-464
        }
    }
    // ---- (5) phase 1 of k2d_uvbar: the shared quantities of this thread's rho and psi point
    const double fac = 1000.0 / 1025.0;
    if (in_rb) {
        gz0 = (fac + rS0) * zw0; gz20 = gz0 * zw0; gSA0 = zw0 * (rS0 - rA0);
        UFx = s2_UFx(ubr, DUon, i, j, P);
        VFe = s2_VFe(vbr, DVom, i, j, P);
        struct C0 { double v; __device__ __forceinline__ double operator()(int, int, int) const { return v; } };
        const MixR r = mix_r_metrics(tpmv, tpnv, C0{vr0}, i, j);
        const double cffr = mix_r_cff(r, Drhs, ubr, vbr, i, j, 0);
        mUFx = r.on_r * r.on_r * r.visc * cffr;
        mVFe = r.om_r * r.om_r * r.visc * cffr;
        if (K2_COR) {
            const double fm = fc0 * (1.0 / (pm0 * pn0));
            const double d = Drhs(i, j, 0);
            corU = 0.5 * d * fm * (vbr(i, j, 0) + vbr(i, j + 1, 0));
            corV = 0.5 * d * fm * (ubr(i, j, 0) + ubr(i + 1, j, 0));
        }
    }
    // mskp * 0.125 (mix_p_metrics) must not be formed where mskp is loaded: hipcc hoists the product to the load, and its wait for that
    // one value (vmcnt(0) in the middle of the load phase) held back every load after it when the masks are read
    asm volatile("" : "+v"(mp0));
    if (in_un) {
        UFe = s2_UFe(ubr, DVom, i, j, P);
        VFx = s2_VFx(vbr, DUon, i, j, P);
        struct C0 { double v; __device__ __forceinline__ double operator()(int, int, int) const { return v; } };
        const MixP p = mix_p_metrics(tpmv, tpnv, C0{vp0}, C0{mp0}, i, j);
        const double cffp = mix_p_cff(p, Drhs, ubr, vbr, i, j, 0);
        mUFe = p.om_p * p.om_p * p.visc * cffp;
        mVFx = p.on_p * p.on_p * p.visc * cffp;
    }
    }
    double *q = lds + tid;
    __syncthreads();                                        // every thread is done with the tiles: the slots may overwrite them
    q[Q_UFX * NT] = UFx; q[Q_VFE * NT] = VFe; q[Q_MUFX * NT] = mUFx; q[Q_MVFE * NT] = mVFe; q[Q_CORU * NT] = corU; q[Q_CORV * NT] = corV;
    q[Q_GZ * NT] = gz0; q[Q_GZ2 * NT] = gz20; q[Q_GSA * NT] = gSA0; q[Q_H * NT] = h0; q[Q_ZW * NT] = zw0; q[Q_RA * NT] = rA0;
    q[Q_PM * NT] = pm0; q[Q_PN * NT] = pn0; q[Q_UFE * NT] = UFe; q[Q_VFX * NT] = VFx; q[Q_MUFE * NT] = mUFe; q[Q_MVFX * NT] = mVFx;
    q[Q_ZN * NT] = zn0;
    __syncthreads();
    // ---- (6) phase 2 of k2d_uvbar (scalars read again, as above)
    {
    const K2Args &L = k2_args_again();
    const UVPar &P = L.P;
    const V2 ub_new{L.A.u2, L.A.U[FA_NEW]}, rubar_rhs{L.A.u2, L.A.U[FA_RRHS]}, rufrc{L.A.u2, L.A.U[FA_FRC]}, ru2d0{L.A.u2, L.A.U[FA_R2D0]}, ru2d1{L.A.u2, L.A.U[FA_R2D1]};
    const V2 vb_new{L.A.v2, L.A.V[FA_NEW]}, rvbar_rhs{L.A.v2, L.A.V[FA_RRHS]}, rvfrc{L.A.v2, L.A.V[FA_FRC]}, rv2d0{L.A.v2, L.A.V[FA_R2D0]}, rv2d1{L.A.v2, L.A.V[FA_R2D1]};
    const double cffg = 0.5 * P.g, third = 1.0 / 3.0;
    if (is_u) {
        const double *w = q - 1, *n = q + RT_BX;
        const double hx = w[Q_H * NT], zwx = w[Q_ZW * NT], rAx = w[Q_RA * NT], pmx = w[Q_PM * NT], pnx = w[Q_PN * NT];
        const double on_u = 2.0 / (pn0 + pnx);
        double rhs = cffg * on_u *
            ((hx + h0) * (w[Q_GZ * NT] - gz0) +
             (hx - h0) * (w[Q_GSA * NT] + gSA0 + third * (rAx - rA0) * (zwx - zw0)) +
             (w[Q_GZ2 * NT] - gz20));
        {
            const double cff1 = UFx - w[Q_UFX * NT];
            const double cff2 = n[Q_UFE * NT] - UFe;
            rhs -= (cff1 + cff2);
        }
        if (K2_COR) rhs += 0.5 * (corU + w[Q_CORU * NT]);
        {
            const double cff1 = 0.5 * (pnx + pn0) * (mUFx - w[Q_MUFX * NT]);
            const double cff2 = 0.5 * (pmx + pm0) * (n[Q_MUFE * NT] - mUFe);
            rhs += cff1 + cff2;
        }
        if (K2_COUPLE == 0) rhs += u_frc;
        else {
            double f = u_frc;
            if (K2_COUPLE == 1) { f -= rhs; rhs += f; ru2d0(i, j, 0) = f; }
            else if (K2_COUPLE == 2) {
                f = f - rhs;
                const double r0 = u_r0;
                rhs = rhs + 1.5 * f - 0.5 * r0;
                ru2d1(i, j, 0) = r0; ru2d0(i, j, 0) = f;
            } else {
                const double c1 = 23.0 / 12.0, c2 = 16.0 / 12.0, c3 = 5.0 / 12.0;
                f = f - rhs;
                const double r0 = u_r0, r1 = u_r1;
                rhs = rhs + c1 * f - c2 * r0 + c3 * r1;
                ru2d1(i, j, 0) = r0; ru2d0(i, j, 0) = f;
            }
            rufrc(i, j, 0) = f;
        }
        const double Dstp0 = zs0 + h0, Dstp1 = u_zsw + hx;
        const double cff = (pm0 + pmx) * (pn0 + pnx);
        const double Dnew_avg = 1.0 / ((zn0 + h0) + (w[Q_ZN * NT] + hx));
        const double mu = u_msk;
        double ubv;
        if (!K2_AM3) ubv = (u_stp * (Dstp0 + Dstp1) + cff * P.c1 * rhs) * Dnew_avg * mu;
        else ubv = (u_stp * (Dstp0 + Dstp1) +
                    cff * (P.c1 * rhs + P.c2 * u_rs - P.c3 * u_rp)) * Dnew_avg * mu;
        if (L.wallbc) {
            // REMORA_BoundaryConditions_xvel.cpp:56-265 for walls on all sides, as its passes leave ubar(knew): ext_dir on the wall
            // face and the two ghost faces behind it; foextrap (x msku) into the two ghost rows; 0.5 * (side + row) in the corners
            const V2 msku{L.A.smu, L.A.msku};
            auto MU = [&](int ii, int jj) -> double { return K2_PNOMASK ? 1.0 : msku(ii, jj, 0); };
            const bool xlo = (i == P.dlox), xhi = (i == P.dhix + 1);
            if (xlo || xhi) {
                const double ext = xlo ? L.bc_ext[0] : L.bc_ext[1]; const int sgn = xlo ? -1 : 1;
                ubv = ext * MU(i, j);
                ub_new(i + sgn, j, 0) = ext * MU(i + sgn, j); ub_new(i + 2 * sgn, j, 0) = ext * MU(i + 2 * sgn, j);
            } else if (j == P.dloy || j == P.dhiy) {
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    if (side == 0 ? (j != P.dloy) : (j != P.dhiy)) continue;
                    const int sgn = side == 0 ? -1 : 1;
                    for (int g = 1; g <= 2; g++) ub_new(i, j + sgn * g, 0) = ubv * MU(i, j + sgn * g);
                    // the corner blocks hang on the first / last interior face of the row
                    if (i == P.dlox + 1)
                        for (int g = 1; g <= 2; g++) for (int ic = P.dlox - 2; ic <= P.dlox; ic++)
                            ub_new(ic, j + sgn * g, 0) = 0.5 * (L.bc_ext[0] * MU(ic, j) + ubv * MU(i, j + sgn * g));
                    if (i == P.dhix)
                        for (int g = 1; g <= 2; g++) for (int ic = P.dhix + 1; ic <= P.dhix + 3; ic++)
                            ub_new(ic, j + sgn * g, 0) = 0.5 * (L.bc_ext[1] * MU(ic, j) + ubv * MU(i, j + sgn * g));
                }
            }
        }
        ub_new(i, j, 0) = ubv;
        if (K2_PPRED) rubar_rhs(i, j, 0) = rhs;
    }
    if (is_v) {
        const double *sp = q - RT_BX, *e = q + 1;
        const double hy = sp[Q_H * NT], zwy = sp[Q_ZW * NT], rAy = sp[Q_RA * NT], pmy = sp[Q_PM * NT], pny = sp[Q_PN * NT];
        const double om_v = 2.0 / (pm0 + pmy);
        double rhs = cffg * om_v *
            ((hy + h0) * (sp[Q_GZ * NT] - gz0) +
             (hy - h0) * (sp[Q_GSA * NT] + gSA0 + third * (rAy - rA0) * (zwy - zw0)) +
             (sp[Q_GZ2 * NT] - gz20));
        {
            const double cff1 = e[Q_VFX * NT] - VFx;
            const double cff2 = VFe - sp[Q_VFE * NT];
            rhs -= (cff1 + cff2);
        }
        if (K2_COR) rhs -= 0.5 * (corV + sp[Q_CORV * NT]);
        {
            const double cff1 = 0.5 * (pny + pn0) * (e[Q_MVFX * NT] - mVFx);
            const double cff2 = 0.5 * (pmy + pm0) * (mVFe - sp[Q_MVFE * NT]);
            rhs += cff1 - cff2;
        }
        if (K2_COUPLE == 0) rhs += v_frc;
        else {
            double f = v_frc;
            if (K2_COUPLE == 1) { f -= rhs; rhs += f; rv2d0(i, j, 0) = f; }
            else if (K2_COUPLE == 2) {
                f = f - rhs;
                const double r0 = v_r0;
                rhs = rhs + 1.5 * f - 0.5 * r0;
                rv2d1(i, j, 0) = r0; rv2d0(i, j, 0) = f;
            } else {
                const double c1 = 23.0 / 12.0, c2 = 16.0 / 12.0, c3 = 5.0 / 12.0;
                f = f - rhs;
                const double r0 = v_r0, r1 = v_r1;
                rhs = rhs + c1 * f - c2 * r0 + c3 * r1;
                rv2d1(i, j, 0) = r0; rv2d0(i, j, 0) = f;
            }
            rvfrc(i, j, 0) = f;
        }
        const double Dstp0 = zs0 + h0, Dstp1 = v_zss + hy;
        const double cff = (pm0 + pmy) * (pn0 + pny);
        const double Dnew_avg = 1.0 / ((zn0 + h0) + (sp[Q_ZN * NT] + hy));
        const double mv = v_msk;
        double vbv;
        if (!K2_AM3) vbv = (v_stp * (Dstp0 + Dstp1) + cff * P.c1 * rhs) * Dnew_avg * mv;
        else vbv = (v_stp * (Dstp0 + Dstp1) +
                    cff * (P.c1 * rhs + P.c2 * v_rs - P.c3 * v_rp)) * Dnew_avg * mv;
        if (L.wallbc) {                                      // REMORA_BoundaryConditions_yvel.cpp:55-265, the mirror image
            const V2 mskv{L.A.smv, L.A.mskv};
            auto MV = [&](int ii, int jj) -> double { return K2_PNOMASK ? 1.0 : mskv(ii, jj, 0); };
            const bool ylo = (j == P.dloy), yhi = (j == P.dhiy + 1);
            if (ylo || yhi) {
                const double ext = ylo ? L.bc_ext[2] : L.bc_ext[3]; const int sgn = ylo ? -1 : 1;
                vbv = ext * MV(i, j);
                vb_new(i, j + sgn, 0) = ext * MV(i, j + sgn); vb_new(i, j + 2 * sgn, 0) = ext * MV(i, j + 2 * sgn);
            } else if (i == P.dlox || i == P.dhix) {
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    if (side == 0 ? (i != P.dlox) : (i != P.dhix)) continue;
                    const int sgn = side == 0 ? -1 : 1;
                    for (int g = 1; g <= 2; g++) vb_new(i + sgn * g, j, 0) = vbv * MV(i + sgn * g, j);
                    if (j == P.dloy + 1)
                        for (int g = 1; g <= 2; g++) for (int jc = P.dloy - 2; jc <= P.dloy; jc++)
                            vb_new(i + sgn * g, jc, 0) = 0.5 * (vbv * MV(i + sgn * g, j) + L.bc_ext[2] * MV(i, jc));
                    if (j == P.dhiy)
                        for (int g = 1; g <= 2; g++) for (int jc = P.dhiy + 1; jc <= P.dhiy + 3; jc++)
                            vb_new(i + sgn * g, jc, 0) = 0.5 * (vbv * MV(i + sgn * g, j) + L.bc_ext[3] * MV(i, jc));
                }
            }
        }
        vb_new(i, j, 0) = vbv;
        if (K2_PPRED) rvbar_rhs(i, j, 0) = rhs;
    }
    }
}
#undef K2_ZNOMASK
#undef K2_PNOMASK
#undef K2_COR
#undef K2_AVG
#undef K2_ZMODE
#undef K2_ZPRED
#undef K2_COUPLE
#undef K2_AM3
#undef K2_PPRED

// k2d_flux on the rim only: the points of tbxp3 outside `hole` (the valid box shrunk by 3) -- every ghost face and every face a
// periodic wrap or a neighbour rank takes its ghost values from.  Same enumeration as k2d_frame.
__global__ void k2d_flux_frame(B3 g3, B3 hole, B3 xg, B3 yg, A4 zeta, A4 h, A4 ubar, A4 vbar, A4 pm, A4 pn, A4 DUon, A4 DVom)
{
    const int NX = g3.hi0 - g3.lo0 + 1, NY = g3.hi1 - g3.lo1 + 1, vx = hole.hi0 - hole.lo0 + 1, vy = hole.hi1 - hole.lo1 + 1;
    const int nglo_x = hole.lo0 - g3.lo0, nglo_y = hole.lo1 - g3.lo1;
    const int nrow_part = (NY - vy) * NX, total = nrow_part + vy * (NX - vx);
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    int i, j;
    if (g < nrow_part) { const int row = g / NX; i = g3.lo0 + g % NX; j = row < nglo_y ? g3.lo1 + row : hole.hi1 + 1 + (row - nglo_y); }
    else { const int r = g - nrow_part, w = NX - vx, row = r / w, cc = r % w; j = hole.lo1 + row; i = cc < nglo_x ? g3.lo0 + cc : hole.hi0 + 1 + (cc - nglo_x); }
    const double d0 = zeta(i, j, 0) + h(i, j, 0);
    if (inb2(xg, i, j)) {
        const double on_u = 2.0 / (pn(i, j, 0) + pn(i - 1, j, 0));
        const double cff1 = 0.5 * on_u * (d0 + (zeta(i - 1, j, 0) + h(i - 1, j, 0)));
        DUon(i, j, 0) = ubar(i, j, 0) * cff1;
    }
    if (inb2(yg, i, j)) {
        const double om_v = 2.0 / (pm(i, j, 0) + pm(i, j - 1, 0));
        const double cff1 = 0.5 * om_v * (d0 + (zeta(i, j - 1, 0) + h(i, j - 1, 0)));
        DVom(i, j, 0) = vbar(i, j, 0) * cff1;
    }
}

// Time averages on the ghost frame: the points of tbxp3 / xgbx2 / ygbx2 outside tbxp1 (k2d_fused owns tbxp1).  One thread per frame
// point: rows below and above tbxp1 at full width, then the side segments of its rows.  DUon / DVom are formed on the fly.
__global__ void k2d_frame(B3 g3, B3 xg, B3 yg, B3 t1, ZetaPar zp, A4 zeta_rhs, A4 h, A4 ub_rhs, A4 vb_rhs, A4 pm, A4 pn,
                          A4 Zt_avg1, A4 DU_avg1, A4 DU_avg2, A4 DV_avg1, A4 DV_avg2, A4 DUon_g, A4 DVom_g, int ghost_from_mem)
{
    const int NX = g3.hi0 - g3.lo0 + 1, NY = g3.hi1 - g3.lo1 + 1, vx = t1.hi0 - t1.lo0 + 1, vy = t1.hi1 - t1.lo1 + 1;
    const int nglo_x = t1.lo0 - g3.lo0, nglo_y = t1.lo1 - g3.lo1;
    const int nrow_part = (NY - vy) * NX, total = nrow_part + vy * (NX - vx);
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    int i, j;
    if (g < nrow_part) { const int row = g / NX; i = g3.lo0 + g % NX; j = row < nglo_y ? g3.lo1 + row : t1.hi1 + 1 + (row - nglo_y); }
    else { const int r = g - nrow_part, w = NX - vx, row = r / w, cc = r % w; j = t1.lo1 + row; i = cc < nglo_x ? g3.lo0 + cc : t1.hi0 + 1 + (cc - nglo_x); }
    const double d0 = zeta_rhs(i, j, 0) + h(i, j, 0);
    const bool inx = inb2(xg, i, j), iny = inb2(yg, i, j);
    double du = 0.0, dv = 0.0;
    if (inx) {                                             // every frame point is a ghost face of DUon and of DVom
        if (ghost_from_mem) du = DUon_g(i, j, 0);
        else {
            const double on_u = 2.0 / (pn(i, j, 0) + pn(i - 1, j, 0));
            const double cff1 = 0.5 * on_u * (d0 + (zeta_rhs(i - 1, j, 0) + h(i - 1, j, 0)));
            du = ub_rhs(i, j, 0) * cff1;
        }
    }
    if (iny) {
        if (ghost_from_mem) dv = DVom_g(i, j, 0);
        else {
            const double om_v = 2.0 / (pm(i, j, 0) + pm(i, j - 1, 0));
            const double cff1 = 0.5 * om_v * (d0 + (zeta_rhs(i, j - 1, 0) + h(i, j - 1, 0)));
            dv = vb_rhs(i, j, 0) * cff1;
        }
    }
    if (zp.avg_mode == 0) {
        Zt_avg1(i, j, 0) = 0.0;
        if (inx) { DU_avg1(i, j, 0) = 0.0; DU_avg2(i, j, 0) = zp.wt_b * du; }
        if (iny) { DV_avg1(i, j, 0) = 0.0; DV_avg2(i, j, 0) = zp.wt_b * dv; }
    } else if (zp.avg_mode == 1) {
        Zt_avg1(i, j, 0) += zp.wt_a * zeta_rhs(i, j, 0);
        if (inx) { DU_avg1(i, j, 0) += zp.wt_a * du; DU_avg2(i, j, 0) += zp.wt_b * du; }
        if (iny) { DV_avg1(i, j, 0) += zp.wt_a * dv; DV_avg2(i, j, 0) += zp.wt_b * dv; }
    } else {
        if (inx) DU_avg2(i, j, 0) = DU_avg2(i, j, 0) + zp.wt_b * du;
        if (iny) DV_avg2(i, j, 0) = DV_avg2(i, j, 0) + zp.wt_b * dv;
    }
}

// REMORA::rhs_uv_2d as a leaf of its own (REMORA_rhs_uv_2d.cpp:20-238): rhs_ubar, rhs_vbar -= advection
__global__ void k_rhs_uv_2d(B3 xb, B3 yb, B3 un, UVPar P, A4 ub, A4 vb, A4 rhs_u, A4 rhs_v, A4 DUon, A4 DVom)
{
    RH_IJ(un);
    if (inb2(xb, i, j)) {
        const double cff1 = s2_UFx(ub, DUon, i, j, P) - s2_UFx(ub, DUon, i - 1, j, P);
        const double cff2 = s2_UFe(ub, DVom, i, j + 1, P) - s2_UFe(ub, DVom, i, j, P);
        rhs_u(i, j, 0) -= (cff1 + cff2);
    }
    if (inb2(yb, i, j)) {
        const double cff1 = s2_VFx(vb, DUon, i + 1, j, P) - s2_VFx(vb, DUon, i, j, P);
        const double cff2 = s2_VFe(vb, DVom, i, j, P) - s2_VFe(vb, DVom, i, j - 1, P);
        rhs_v(i, j, 0) -= (cff1 + cff2);
    }
}
static UVPar uvpar_base(const rh_ctx *c)
{
    UVPar P; memset(&P, 0, sizeof P);
    P.dlox = c->dom.lo[0]; P.dhix = c->dom.hi[0]; P.dloy = c->dom.lo[1]; P.dhiy = c->dom.hi[1];
    P.perx = c->prm.periodic[0]; P.pery = c->prm.periodic[1]; P.uv_c2 = c->prm.uv_c2; P.use_cor = c->prm.use_coriolis; P.g = c->prm.g;
    P.nomask = c->prm.masks_all_one ? 1 : 0;
    return P;
}
static int need_uv2d_stencil(const rh_box &xb, const rh_box &yb, const rh_array4 &ub, const rh_array4 &vb, const rh_array4 &DUon, const rh_array4 &DVom)
{
    rh_box q = bx_grow(xb, 2, 2, 0); RH_NEED(ub, q); q = bx_grow(yb, 2, 2, 0); RH_NEED(vb, q);
    q = xb; q.lo[0] -= 2; q.hi[0] += 2; RH_NEED(DUon, q); q = xb; q.lo[0] -= 2; q.hi[0] += 1; q.hi[1] += 1; RH_NEED(DVom, q);
    q = yb; q.lo[1] -= 2; q.hi[1] += 2; RH_NEED(DVom, q); q = yb; q.lo[1] -= 2; q.hi[1] += 1; q.hi[0] += 1; RH_NEED(DUon, q);
    return 0;
}
extern "C" int rh_rhs_uv_2d(rh_ctx *c, rh_box xbx, rh_box ybx, rh_array4 ubar, rh_array4 vbar, rh_array4 rhs_ubar, rh_array4 rhs_vbar,
                            rh_array4 DUon, rh_array4 DVom, int krhs)
{
    TScope t(c, "rhs_uv_2d");
    xbx = bx_slab(xbx, 2, 0); ybx = bx_slab(ybx, 2, 0);
    if (krhs < 0 || krhs >= ubar.ncomp || krhs >= vbar.ncomp) return rh_fail("rh_rhs_uv_2d: krhs out of range");
    const rh_array4 ub = a4_comp(ubar, krhs), vb = a4_comp(vbar, krhs);
    RH_TRY(need_uv2d_stencil(xbx, ybx, ub, vb, DUon, DVom));
    RH_NEED(rhs_ubar, xbx); RH_NEED(rhs_vbar, ybx);
    const rh_box un = bx_union(xbx, ybx);
    k_rhs_uv_2d<<<grid2(un), rh_block(), 0, c->stream>>>(mkB(xbx), mkB(ybx), mkB(un), uvpar_base(c), mkA(ub), mkA(vb), mkA(rhs_ubar), mkA(rhs_vbar), mkA(DUon), mkA(DVom));
    RH_LAUNCH_CHECK(); return 0;
}

// ------------------------------------------------------------------------------------------------------------------
extern "C" int rh_advance_2d(rh_ctx *c, rh_state *s, int istep, double dtfast, int predictor, int first, int my_iif, int *next_indx1)
{
    TScope tsc(c, "advance_2d");
    const int iic = istep, ntfirst = 0;
    // time indices, :79-101
    int knew = 3;
    int krhs = (my_iif + iic) % 2 + 1;
    int kstp = my_iif <= 1 ? iic % 2 + 1 : (iic % 2 + my_iif % 2 + 1) % 2 + 1;
    const int indx1 = krhs;
    if (predictor) *next_indx1 = 3 - indx1;
    else { knew = *next_indx1; kstp = 3 - knew; krhs = 3; }
    int ptsk = 3 - kstp;
    knew -= 1; krhs -= 1; kstp -= 1; ptsk -= 1;
    if (knew < 0 || knew > 2 || kstp < 0 || kstp > 1 || ptsk < 0 || ptsk > 1) return rh_fail("rh_advance_2d: inconsistent time indices (next_indx1 = %d)", *next_indx1);
    const int nfast = c->nfast;
    if (my_iif < 0 || my_iif + 1 > 2 * c->prm.ndtfast) return rh_fail("rh_advance_2d: my_iif out of range");
    const double *w1 = c->w1.data(), *w2 = c->w2.data();

    const rh_box b2 = bx_slab(c->bx, 2, 0);
    const rh_box tbxp1 = bx_grow(b2, 1, 1, 0), tbxp3 = bx_grow(b2, 3, 3, 0);
    const rh_box xgbx2 = bx_grow(bx_nodal(b2, 0), 2, 2, 0), ygbx2 = bx_grow(bx_nodal(b2, 1), 2, 2, 0);
    const rh_box xbxD = bx_nodal(b2, 0), ybxD = bx_nodal(b2, 1);
    if (s->zeta.ncomp < 3 || s->ubar.ncomp < 3 || s->vbar.ncomp < 3 || s->rzeta.ncomp < 2 || s->rubar.ncomp < 3 || s->rvbar.ncomp < 3 ||
        s->ru2d.ncomp < 2 || s->rv2d.ncomp < 2 || s->h.ncomp < 1)
        return rh_fail("rh_advance_2d: a barotropic array has too few components");
    if (!predictor && s->rzeta.ncomp < 2) return rh_fail("rh_advance_2d: rzeta");

    const rh_array4 zeta_rhs = a4_comp(s->zeta, krhs), zeta_stp = a4_comp(s->zeta, kstp), zeta_new = a4_comp(s->zeta, knew);
    const rh_array4 ub_rhs = a4_comp(s->ubar, krhs), vb_rhs = a4_comp(s->vbar, krhs);

    // k2d_fused takes one shape per group of arrays (see K2Arrays): native states and REMORA's MultiFabs have them by construction
    auto same2 = [](std::initializer_list<const rh_array4 *> l) {
        const rh_array4 *f = *l.begin();
        for (const rh_array4 *a : l) if (a->lo[0] != f->lo[0] || a->lo[1] != f->lo[1] || a->hi[0] != f->hi[0] || a->hi[1] != f->hi[1] || a->lo[2] != 0) return false;
        return true;
    };
    const bool shapes_ok = same2({&s->zeta, &s->h, &s->fcor, &s->mskr, &s->Zt_avg1}) && same2({&s->rzeta, &s->rhoS, &s->rhoA, &s->visc2_r, &s->visc2_p}) &&
                           same2({&s->ubar, &s->rubar, &s->rufrc, &s->ru2d, &s->DU_avg1, &s->DU_avg2, &c->DUon}) &&
                           same2({&s->vbar, &s->rvbar, &s->rvfrc, &s->rv2d, &s->DV_avg1, &s->DV_avg2, &c->DVom});
    const bool fused = !c->unfused && !c->nofuse2d && my_iif < nfast && shapes_ok;   // the last call only accumulates the averages (:362-363): three-kernel form
    // Walls on all four sides of a one-rank domain (e.g. Exec/DoubleGyre): FillPatch of the new level is local to the rim, and the
    // fused kernel's rim blocks do it (three to four launches fewer per call).  Everything else takes the batched BC passes.
    auto wall = [&](int v, int dn, int dt_) {              // variable v: ext_dir in its normal direction dn, foextrap in the other
        return c->bcs[v].lo[dn] == RH_BC_EXT_DIR && c->bcs[v].hi[dn] == RH_BC_EXT_DIR && c->bcs[v].lo[dt_] == RH_BC_FOEXTRAP && c->bcs[v].hi[dt_] == RH_BC_FOEXTRAP; };
    const bool wallbc = fused && !c->no_bc2d && !c->prm.periodic[0] && !c->prm.periodic[1] && c->prm.nranks[0] <= 1 && c->prm.nranks[1] <= 1 &&
        b2.hi[0] - b2.lo[0] >= 2 && b2.hi[1] - b2.lo[1] >= 2 && wall(RH_BCV_UBAR, 0, 1) && wall(RH_BCV_VBAR, 1, 0) &&
        c->bcs[RH_BCV_ZETA].lo[0] == RH_BC_FOEXTRAP && c->bcs[RH_BCV_ZETA].hi[0] == RH_BC_FOEXTRAP && c->bcs[RH_BCV_ZETA].lo[1] == RH_BC_FOEXTRAP && c->bcs[RH_BCV_ZETA].hi[1] == RH_BC_FOEXTRAP &&
        s->zeta.lo[0] == b2.lo[0] - 3 && s->zeta.hi[0] == b2.hi[0] + 3 && s->zeta.lo[1] == b2.lo[1] - 3 && s->zeta.hi[1] == b2.hi[1] + 3 &&
        s->ubar.lo[0] == b2.lo[0] - 2 && s->ubar.hi[0] == b2.hi[0] + 3 && s->ubar.lo[1] == b2.lo[1] - 2 && s->ubar.hi[1] == b2.hi[1] + 2 &&
        s->vbar.lo[0] == b2.lo[0] - 2 && s->vbar.hi[0] == b2.hi[0] + 2 && s->vbar.lo[1] == b2.lo[1] - 2 && s->vbar.hi[1] == b2.hi[1] + 3;
    // ---- Drhs, DUon, DVom (:149-166) and their ghost exchange (:170-171)
    RH_NEED(s->zeta, tbxp3); RH_NEED(s->h, tbxp3);
    RH_NEED(s->ubar, xgbx2); RH_NEED(s->vbar, ygbx2); RH_NEED(c->DUon, xgbx2); RH_NEED(c->DVom, ygbx2);
    RH_NEED(s->pn, bx_growlo_i(xgbx2)); RH_NEED(s->pm, bx_growlo_j(ygbx2));
    // FillBoundary of DUon / DVom has work to do only in a periodic direction or next to a neighbour rank
    const bool fb_work = c->prm.periodic[0] || c->prm.periodic[1] || c->prm.nranks[0] > 1 || c->prm.nranks[1] > 1;
    const rh_box hole = bx_grow(b2, -3, -3, 0);
    auto flux_exchange = [&]() -> int {
    if (!fused || fb_work) {
        if (!fused || !bx_ok(hole))
            k2d_flux<<<grid2(tbxp3), rh_block(), 0, c->stream>>>(mkB(tbxp3), mkB(xgbx2), mkB(ygbx2), mkA(zeta_rhs), mkA(s->h), mkA(ub_rhs), mkA(vb_rhs),
                mkA(s->pm), mkA(s->pn), mkA(c->DUon), mkA(c->DVom));
        else {
            const int nfr = (int)(bx_npts(tbxp3) - bx_npts(hole));
            k2d_flux_frame<<<(nfr + 255) / 256, 256, 0, c->stream>>>(mkB(tbxp3), mkB(hole), mkB(xgbx2), mkB(ygbx2), mkA(zeta_rhs), mkA(s->h), mkA(ub_rhs), mkA(vb_rhs),
                mkA(s->pm), mkA(s->pn), mkA(c->DUon), mkA(c->DVom));
        }
        RH_LAUNCH_CHECK();
        BcScope xs(c);                                                         // both arrays share one message per neighbour
        RH_TRY(halo_fill_boundary(c, mfv(c, c->DUon, 1, 0, 0), 0, 1));
        RH_TRY(halo_fill_boundary(c, mfv(c, c->DVom, 0, 1, 0), 0, 1));
        RH_TRY(xs.end());
    }
    return 0; };
    // Decomposed run inside rh_advance's barotropic loop: the interior tiles of k2d_fused go to the compute stream first; the side
    // stream then runs the ghost fill the previous call left owed, this call's flux frame and the DUon / DVom exchange under them;
    // the rim tiles wait for that; the new level's ghost fill is left owed to the next call (or to the join in rh_advance_3d_ml).
    bool split = fused && c->in_loop2d && !c->no_async && !c->no_split2d && (c->prm.nranks[0] > 1 || c->prm.nranks[1] > 1) &&
                 (c->xfn || (c->comm && c->async_rccl)) && bx_ok(hole);
    if (split) {                                             // at least one interior tile, or the split only costs a launch
        const rh_box un0 = bx_union(xbxD, ybxD);
        const int ntx = (un0.hi[0] - un0.lo[0] + U_SX) / U_SX, nty = (un0.hi[1] - un0.lo[1] + U_SY) / U_SY;
        const B2v vb0{b2.lo[0], b2.lo[1], b2.hi[0], b2.hi[1]};
        split = ntx >= 3 && nty >= 3 && k2_tile_is_interior(vb0, un0.lo[0] - 1 + (ntx / 2) * U_SX, un0.lo[1] - 1 + (nty / 2) * U_SY);
    }
    if (!split) { RH_TRY(step2d_flush_pending(c)); RH_TRY(flux_exchange()); }

    // ---- averages + free surface
    ZetaPar zp; memset(&zp, 0, sizeof zp);
    zp.predictor = predictor ? 1 : 0; zp.dtfast = dtfast; zp.do_step = my_iif < nfast; zp.nomask = c->prm.masks_all_one ? 1 : 0;
    if (predictor) {
        if (first) { zp.avg_mode = 0; zp.wt_b = (-1.0 / 12.0) * w2[my_iif + 1]; }
        else {
            if (my_iif < 1) return rh_fail("rh_advance_2d: first_2d_step must be set for my_iif == 0");
            zp.avg_mode = 1; zp.wt_a = w1[my_iif - 1]; zp.wt_b = (8.0 / 12.0) * w2[my_iif] - (1.0 / 12.0) * w2[my_iif + 1];
        }
    } else { zp.avg_mode = 2; zp.wt_b = first ? w2[my_iif] : 5.0 / 12.0 * w2[my_iif]; }
    zp.zmode = (my_iif == 0) ? 0 : (predictor ? 1 : 2);
    RH_NEED(s->Zt_avg1, tbxp3); RH_NEED(s->DU_avg1, xgbx2); RH_NEED(s->DU_avg2, xgbx2); RH_NEED(s->DV_avg1, ygbx2); RH_NEED(s->DV_avg2, ygbx2);
    RH_NEED(s->rzeta, tbxp1); RH_NEED(s->pm, tbxp1); RH_NEED(s->pn, tbxp1); RH_NEED(s->mskr, tbxp1); RH_NEED(s->rhoS, tbxp1); RH_NEED(s->rhoA, tbxp1);
    RH_NEED(c->zwrk, tbxp1);
    if (!fused)
    k2d_zeta<<<grid2(tbxp3), rh_block(), 0, c->stream>>>(mkB(tbxp3), mkB(xgbx2), mkB(ygbx2), mkB(tbxp1), zp, mkA(zeta_rhs), mkA(zeta_stp), mkA(zeta_new),
        mkA(a4_comp(s->rzeta, kstp)), mkA(a4_comp(s->rzeta, ptsk)), mkA(a4_comp(s->rzeta, krhs < 2 ? krhs : 0)), mkA(s->h), mkA(s->pm), mkA(s->pn),
        mkA(s->mskr), mkA(s->rhoS), mkA(s->rhoA), mkA(c->DUon), mkA(c->DVom), mkA(s->Zt_avg1), mkA(s->DU_avg1), mkA(s->DU_avg2), mkA(s->DV_avg1), mkA(s->DV_avg2),
        mkA(c->zwrk));
    RH_LAUNCH_CHECK();
    if (my_iif >= nfast) return 0;                                             // :362-363

    // ---- momentum
    UVPar P = uvpar_base(c);
    P.predictor = predictor ? 1 : 0;
    if (first && predictor) P.couple = (iic == ntfirst) ? 1 : (iic == ntfirst + 1 ? 2 : 3); else P.couple = 0;
    if (my_iif == 0 || predictor) { P.am3 = 0; P.c1 = (my_iif == 0) ? 0.5 * dtfast : dtfast; }
    else { P.am3 = 1; P.c1 = 0.5 * dtfast * 5.0 / 12.0; P.c2 = 0.5 * dtfast * 8.0 / 12.0; P.c3 = 0.5 * dtfast * 1.0 / 12.0; }
    RH_TRY(need_uv2d_stencil(xbxD, ybxD, ub_rhs, vb_rhs, c->DUon, c->DVom));
    {
        const rh_box un = bx_union(xbxD, ybxD), g1 = bx_grow(un, 1, 1, 0);
        RH_NEED(s->pm, bx_grow(tbxp1, 1, 1, 0)); RH_NEED(s->pn, bx_grow(tbxp1, 1, 1, 0)); RH_NEED(s->visc2_r, tbxp1); RH_NEED(s->visc2_p, un); RH_NEED(s->mskp, un);
        RH_NEED(s->h, g1); RH_NEED(s->zeta, g1); RH_NEED(s->rhoA, tbxp1); RH_NEED(s->rhoS, tbxp1); RH_NEED(c->zwrk, tbxp1);
        if (c->prm.use_coriolis) RH_NEED(s->fcor, tbxp1);
        // the rho-point fluxes are evaluated on tbxp1 (cells lo-1 .. hi+1): their stencils must stay inside the arrays
        { rh_box q = tbxp1; q.lo[0] -= 1; q.hi[0] += 2; RH_NEED(s->ubar, q); RH_NEED(c->DUon, q); q = tbxp1; q.lo[1] -= 1; q.hi[1] += 2; RH_NEED(s->vbar, q); RH_NEED(c->DVom, q); }
        RH_NEED(s->msku, xbxD); RH_NEED(s->mskv, ybxD); RH_NEED(s->rufrc, xbxD); RH_NEED(s->rvfrc, ybxD); RH_NEED(s->ru2d, xbxD); RH_NEED(s->rv2d, ybxD);
        RH_NEED(s->rubar, xbxD); RH_NEED(s->rvbar, ybxD); RH_NEED(s->ubar, xbxD); RH_NEED(s->vbar, ybxD);
        // rubar/rvbar(krhs) only exist for the predictor (krhs < 2); the corrector never writes them
        const int kr = krhs < 2 ? krhs : 0;
        const dim3 grid((un.hi[0] - un.lo[0] + U_SX) / U_SX, (un.hi[1] - un.lo[1] + U_SY) / U_SY, 1);
        if (fused) {
            // tiles reach 2 points beyond the block's first / one beyond its last point: inside tbxp3 wherever they are consumed
            // the ghost frame's time averages ride on extra rows of the same launch (one thread per frame point)
            const int nfr = (int)(bx_npts(tbxp3) - bx_npts(tbxp1));
            const int nfb = (nfr + RT_BX * U_BY - 1) / (RT_BX * U_BY), frame_rows = (nfb + (int)grid.x - 1) / (int)grid.x;
            K2Arrays KA; memset(&KA, 0, sizeof KA);
            auto shape = [](const rh_array4 &a) { S2 q; q.lo0 = a.lo[0]; q.lo1 = a.lo[1]; q.hi0 = a.hi[0]; q.hi1 = a.hi[1]; q.sy = (unsigned)(a.hi[0] - a.lo[0] + 1); return q; };
            auto comp = [](const rh_array4 &a, int n) { return mkA(a, n).p; };
            KA.r3 = shape(s->zeta); KA.r2 = shape(s->rzeta); KA.u2 = shape(s->ubar); KA.v2 = shape(s->vbar);
            KA.R3[R3_ZRHS] = zeta_rhs.p; KA.R3[R3_ZSTP] = zeta_stp.p; KA.R3[R3_ZNEW] = zeta_new.p; KA.R3[R3_H] = s->h.p; KA.R3[R3_FCOR] = s->fcor.p;
            KA.R3[R3_MSKR] = s->mskr.p; KA.R3[R3_ZT] = s->Zt_avg1.p;
            KA.R2[R2_RZSTP] = comp(s->rzeta, kstp); KA.R2[R2_RZPTS] = comp(s->rzeta, ptsk); KA.R2[R2_RZRHS] = comp(s->rzeta, kr);
            KA.R2[R2_RHOS] = s->rhoS.p; KA.R2[R2_RHOA] = s->rhoA.p; KA.R2[R2_VR] = s->visc2_r.p; KA.R2[R2_VP] = s->visc2_p.p;
            KA.U[FA_RHS] = ub_rhs.p; KA.U[FA_STP] = comp(s->ubar, kstp); KA.U[FA_NEW] = comp(s->ubar, knew); KA.U[FA_RRHS] = comp(s->rubar, kr);
            KA.U[FA_RSTP] = comp(s->rubar, kstp); KA.U[FA_RPTS] = comp(s->rubar, ptsk); KA.U[FA_FRC] = s->rufrc.p; KA.U[FA_R2D0] = comp(s->ru2d, 0);
            KA.U[FA_R2D1] = comp(s->ru2d, 1); KA.U[FA_AVG1] = s->DU_avg1.p; KA.U[FA_AVG2] = s->DU_avg2.p; KA.U[FA_DON] = c->DUon.p;
            KA.V[FA_RHS] = vb_rhs.p; KA.V[FA_STP] = comp(s->vbar, kstp); KA.V[FA_NEW] = comp(s->vbar, knew); KA.V[FA_RRHS] = comp(s->rvbar, kr);
            KA.V[FA_RSTP] = comp(s->rvbar, kstp); KA.V[FA_RPTS] = comp(s->rvbar, ptsk); KA.V[FA_FRC] = s->rvfrc.p; KA.V[FA_R2D0] = comp(s->rv2d, 0);
            KA.V[FA_R2D1] = comp(s->rv2d, 1); KA.V[FA_AVG1] = s->DV_avg1.p; KA.V[FA_AVG2] = s->DV_avg2.p; KA.V[FA_DON] = c->DVom.p;
            KA.spm = shape(s->pm); KA.spn = shape(s->pn); KA.smu = shape(s->msku); KA.smv = shape(s->mskv); KA.smp = shape(s->mskp);
            KA.pm = s->pm.p; KA.pn = s->pn.p; KA.msku = s->msku.p; KA.mskv = s->mskv.p; KA.mskp = s->mskp.p;
            if (s->pm.lo[2] != 0 || s->pn.lo[2] != 0 || s->msku.lo[2] != 0 || s->mskv.lo[2] != 0 || s->mskp.lo[2] != 0) return rh_fail("rh_advance_2d: 2-D arrays must sit on the k = 0 slab");
            const B2v vbx{b2.lo[0], b2.lo[1], b2.hi[0], b2.hi[1]};
            K2Args ka; ka.vb = vbx; ka.P = P; ka.zp = zp; ka.A = KA; ka.ghost_from_mem = fb_work ? 1 : 0; ka.ntile_x = (int)grid.x; ka.ntile_y = (int)grid.y; ka.nframe = nfb;
            ka.wallbc = wallbc ? 1 : 0;
            ka.bc_ext[0] = c->extdir[RH_BCV_UBAR][0]; ka.bc_ext[1] = c->extdir[RH_BCV_UBAR][3]; ka.bc_ext[2] = c->extdir[RH_BCV_VBAR][1]; ka.bc_ext[3] = c->extdir[RH_BCV_VBAR][4];
            ka.part = 0;
            // the kernel's compiled modes (see k2d_fused): steady predictor, steady corrector, or everything from the arguments
            const int kmode = c->no_k2mode ? 0 :
                (zp.avg_mode == 1 && zp.zmode == 1 && zp.predictor == 1 && P.couple == 0 && P.am3 == 0 && P.predictor == 1) ? 1 :
                (zp.avg_mode == 2 && zp.zmode == 2 && zp.predictor == 0 && P.couple == 0 && P.am3 == 1 && P.predictor == 0) ? 2 : 0;
            auto k2_launch = [&](dim3 g, const K2Args &arg) {
                const dim3 b(RT_BX, U_BY, 1);
                const bool fix = zp.nomask && P.nomask && P.use_cor;
                if (kmode == 1) { if (fix) k2d_fused<1, 1><<<g, b, 0, c->stream>>>(arg); else k2d_fused<1, 0><<<g, b, 0, c->stream>>>(arg); }
                else if (kmode == 2) { if (fix) k2d_fused<2, 1><<<g, b, 0, c->stream>>>(arg); else k2d_fused<2, 0><<<g, b, 0, c->stream>>>(arg); }
                else k2d_fused<0, 0><<<g, b, 0, c->stream>>>(arg);
            };
            if (!split) {
                { TScope tk(c, "k2d_fused"); k2_launch(dim3(grid.x, grid.y + (unsigned)frame_rows, 1), ka); }
                RH_LAUNCH_CHECK();
            } else {
                if (!c->ev2d_m) { RH_HIP(hipEventCreateWithFlags(&c->ev2d_m, hipEventDisableTiming)); RH_HIP(hipEventCreateWithFlags(&c->ev2d_b, hipEventDisableTiming)); }
                RH_HIP(hipEventRecord(c->ev2d_m, c->stream));                    // everything up to the previous call's rim tiles
                K2Args ki = ka; ki.part = 1; ki.nframe = 0;
                { TScope tk(c, "k2d_fused"); k2_launch(dim3(grid.x, grid.y, 1), ki); }
                RH_LAUNCH_CHECK();
                RH_TRY(halo_side_enter(c));                                       // c->stream is the side stream from here ...
                int rc = 0;
                if (hipStreamWaitEvent(c->stream, c->ev2d_m, 0) != hipSuccess) rc = rh_fail("rh_advance_2d: hipStreamWaitEvent failed");
                if (!rc) rc = step2d_flush_pending(c);
                if (!rc) rc = flux_exchange();
                if (!rc && hipEventRecord(c->ev2d_b, c->stream) != hipSuccess) rc = rh_fail("rh_advance_2d: hipEventRecord failed");
                const int rc2 = halo_side_leave(c);                               // ... to here
                RH_TRY(rc); RH_TRY(rc2);
                RH_HIP(hipStreamWaitEvent(c->stream, c->ev2d_b, 0));
                ka.part = 2;
                { TScope tk(c, "k2d_fused"); k2_launch(dim3(grid.x, grid.y + (unsigned)frame_rows, 1), ka); }
                RH_LAUNCH_CHECK();
                c->pend2d.on = true; c->pend2d.ubar = s->ubar; c->pend2d.vbar = s->vbar; c->pend2d.zeta = s->zeta;
                c->pend2d.msku = s->msku; c->pend2d.mskv = s->mskv; c->pend2d.mskr = s->mskr; c->pend2d.knew = knew;
                return 0;
            }
        } else
        k2d_uvbar<<<grid, dim3(RT_BX, U_BY, 1), 0, c->stream>>>(mkB(xbxD), mkB(ybxD), mkB(un), mkB(tbxp1), P, mkA(ub_rhs), mkA(vb_rhs),
            mkA(a4_comp(s->ubar, kstp)), mkA(a4_comp(s->vbar, kstp)), mkA(a4_comp(s->ubar, knew)), mkA(a4_comp(s->vbar, knew)),
            mkA(a4_comp(s->rubar, kr)), mkA(a4_comp(s->rvbar, kr)), mkA(a4_comp(s->rubar, kstp)), mkA(a4_comp(s->rvbar, kstp)),
            mkA(a4_comp(s->rubar, ptsk)), mkA(a4_comp(s->rvbar, ptsk)),
            mkA(zeta_rhs), mkA(zeta_stp), mkA(zeta_new), mkA(s->h), mkA(s->pm), mkA(s->pn), mkA(s->fcor), mkA(s->rhoS), mkA(s->rhoA),
            mkA(s->msku), mkA(s->mskv), mkA(s->mskp), mkA(s->visc2_p), mkA(s->visc2_r), mkA(s->rufrc), mkA(s->rvfrc), mkA(s->ru2d), mkA(s->rv2d),
            mkA(c->DUon), mkA(c->DVom), mkA(c->zwrk));
        RH_LAUNCH_CHECK();
    }
    // ---- ghost cells of the new time level, :735-755
    if (wallbc) return 0;                                  // the rim blocks of k2d_fused wrote them
    BcScope bcs(c);
    RH_TRY(halo_fillpatch(c, mfv(c, s->ubar, 1, 0, 0), s->msku, RH_BCV_UBAR, knew, 1));
    RH_TRY(halo_fillpatch(c, mfv(c, s->vbar, 0, 1, 0), s->mskv, RH_BCV_VBAR, knew, 1));
    RH_TRY(halo_fillpatch(c, mfv(c, s->zeta, 0, 0, 0), s->mskr, RH_BCV_ZETA, knew, 1));
    return bcs.end();
}

// the ghost cells of the new time level (:735-755) that a split call left owed; runs on whatever stream the context points at
int step2d_flush_pending(rh_ctx *c)
{
    if (!c->pend2d.on) return 0;
    c->pend2d.on = false;
    const int knew = c->pend2d.knew;
    BcScope bcs(c);
    RH_TRY(halo_fillpatch(c, mfv(c, c->pend2d.ubar, 1, 0, 0), c->pend2d.msku, RH_BCV_UBAR, knew, 1));
    RH_TRY(halo_fillpatch(c, mfv(c, c->pend2d.vbar, 0, 1, 0), c->pend2d.mskv, RH_BCV_VBAR, knew, 1));
    RH_TRY(halo_fillpatch(c, mfv(c, c->pend2d.zeta, 0, 0, 0), c->pend2d.mskr, RH_BCV_ZETA, knew, 1));
    return bcs.end();
}

// REMORA::advance_2d_onestep, REMORA_advance_2d_onestep.cpp:6-53
extern "C" int rh_advance_2d_onestep(rh_ctx *c, rh_state *s, int istep, int my_iif)
{
    const int nfast_counter = c->nfast + 1;
    const double dtfast = c->prm.dt / (double)c->prm.ndtfast;
    const int first = (my_iif == 0);
    int next_indx1 = 0;
    RH_TRY(rh_advance_2d(c, s, istep, dtfast, 1, first, my_iif, &next_indx1));
    if (my_iif < nfast_counter - 1) RH_TRY(rh_advance_2d(c, s, istep, dtfast, 0, first, my_iif, &next_indx1));
    return 0;
}
