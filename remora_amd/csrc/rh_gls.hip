// rh_gls.hip -- generic-length-scale turbulence closure on device (vertical_mixing_type = gls).
//   REMORA::gls_prestep    Source/TimeIntegration/REMORA_gls.cpp:5-216    predictor of tke / gls on w-points
//   REMORA::gls_corrector  Source/TimeIntegration/REMORA_gls.cpp:218-915  advection, production / dissipation, implicit vertical
//                                                                          diffusion, stability functions -> Akv, Akt, Akk, Akp, Lscale
// The reference runs ~25 ParallelFor passes over 20 scratch FABs per call; here: one column kernel for the predictor, and for the
// corrector a shear column kernel, a 4-point averaging pass, a horizontal-advection pass and ONE column kernel that carries the
// vertical advection, production / dissipation, both tridiagonal solves and the stability functions (three w-point scratch arrays).
// Arithmetic keeps the reference's operation order; the k-invariant pow() factors are formed on the host (as the reference does on
// its host), the per-point pow() calls run on device (ocml), so parity with the CPU oracle is to rounding, not bit-exact.
#include "rh_common.h"
#include "rh_tile.h"

// the tiled kernels' blocks (64 x 8 threads) own 63 x 7 columns: they overlap by one column and one row
#define GH_SX (RT_BX - 1)
#define GH_SY (RT_BY - 1)

struct GlsC {
    int stab, Lmy25, N, perx, pery, dlo0, dhi0, dlo1, dhi1;
    double gls_m, gls_n, Kmin, Pmin, cmu0, c1, c2, c3m, c3p, sigk, sigp_cb, ogls_sigp, Akk_bak, Akp_bak, Akv_bak, Akt_bak;
    double Gh0, Ghcri, Ghmin, E2, vonKar, L_sft, sqrt2, cmu_fac1, cmu_fac2, cmu_fac3, fac2, fac3, fac4, fac5, fac6;
    double gls_exp1, tke_exp1, tke_exp2, tke_exp4;
    double s0, s1, s2, s4, s5, s6, b0, b1, b2, b3, b4, b5, Sm2, Sm3, Sm4, Sh1, Sh2;
    double Zos_eff, Zob_min, Gadv, eps;
    double cmu0_p, surf_len_n, bot_len_n, Lsft_n;      // pow(cmu0,p), pow(L_sft*Zos_eff,n), pow(vonKar*Zob_min,n), pow(L_sft,n)
    // exponent classes of the per-point powers (see gpow): m, n, -n, n-1, 1/n, -1/n, -m/n, 0.5+m/n, m+0.5n
    int k_m, k_n, k_mn, k_nm1, k_e1, k_me1, k_mt1, k_t2, k_t4;
};

// The closure raises tke, gls and the length scale to exponents fixed by (gls_p, gls_m, gls_n).  For the named closures these are
// multiples of 1/2 (k-epsilon: -1, -1.5, 1.5, -2, 1; k-omega: -1, -0.5, 0.5, ...), where a correctly rounded division / square
// root / product chain is within one rounding or two of libm's pow() and ~20x cheaper than ocml's double pow (the column kernel
// calls it ten times per point).  Any other exponent (the "gen" closure, n = -0.67) takes pow().
enum { GP_POW = 0, GP_ZERO, GP_ONE, GP_MONE, GP_HALF, GP_MHALF, GP_1P5, GP_M1P5, GP_TWO, GP_MTWO };
static int gpow_class(double e)
{
    if (e == 0.0) return GP_ZERO; if (e == 1.0) return GP_ONE; if (e == -1.0) return GP_MONE; if (e == 0.5) return GP_HALF;
    if (e == -0.5) return GP_MHALF; if (e == 1.5) return GP_1P5; if (e == -1.5) return GP_M1P5; if (e == 2.0) return GP_TWO;
    if (e == -2.0) return GP_MTWO; return GP_POW;
}
__device__ __forceinline__ double gpow(double x, double e, int cls)
{
    switch (cls) {
    case GP_ZERO: return 1.0;
    case GP_ONE: return x;
    case GP_MONE: return 1.0 / x;
    case GP_HALF: return sqrt(x);
    case GP_MHALF: return 1.0 / sqrt(x);
    case GP_1P5: return x * sqrt(x);
    case GP_M1P5: return 1.0 / (x * sqrt(x));
    case GP_TWO: return x * x;
    case GP_MTWO: return 1.0 / (x * x);
    default: return pow(x, e);
    }
}

// REMORA_gls.cpp:230-349 (+ the defaults of REMORA_DataStruct.H:243-268,399-400,417,450-467), all on the host in double
static GlsC gls_constants(const rh_ctx *c)
{
    const rh_gls_params &G = c->prm.gls;
    GlsC q; memset(&q, 0, sizeof q);
    q.stab = G.stability; q.N = c->N; q.perx = c->prm.periodic[0]; q.pery = c->prm.periodic[1];
    q.dlo0 = c->dom.lo[0]; q.dhi0 = c->dom.hi[0]; q.dlo1 = c->dom.lo[1]; q.dhi1 = c->dom.hi[1];
    q.Lmy25 = (G.gls_p == 0.0 && G.gls_n == 1.0 && G.gls_m == 1.0);
    q.vonKar = 0.41; q.L_sft = q.vonKar; q.sigp_cb = G.gls_sigp; q.ogls_sigp = 1.0 / q.sigp_cb;
    q.gls_m = G.gls_m; q.gls_n = G.gls_n; q.Kmin = G.gls_Kmin; q.Pmin = G.gls_Pmin; q.cmu0 = G.gls_cmu0;
    q.c1 = G.gls_c1; q.c2 = G.gls_c2; q.c3m = G.gls_c3m; q.c3p = G.gls_c3p; q.sigk = G.gls_sigk;
    q.Akk_bak = G.Akk_bak; q.Akp_bak = G.Akp_bak; q.Akv_bak = G.Akv_bak; q.Akt_bak = G.Akt_bak;
    q.Ghmin = -0.28; q.E2 = 1.33;
    double L1 = 0, L2 = 0, L3 = 0, L4 = 0, L5 = 0, L6 = 0, L7 = 0, L8 = 0;
    if (G.stability == RH_GLS_CANUTO_A) { q.Gh0 = 0.0329; q.Ghcri = 0.03; L1 = 0.107; L2 = 0.0032; L3 = 0.0864; L4 = 0.12; L5 = 11.9; L6 = 0.4; L7 = 0.0; L8 = 0.48; }
    else if (G.stability == RH_GLS_CANUTO_B) { q.Gh0 = 0.0444; q.Ghcri = 0.0414; L1 = 0.127; L2 = 0.00336; L3 = 0.0906; L4 = 0.101; L5 = 11.2; L6 = 0.4; L7 = 0.0; L8 = 0.318; }
    else { q.Gh0 = 0.028; q.Ghcri = 0.02; }
    const double p = G.gls_p, n = G.gls_n, m = G.gls_m, cmu0 = G.gls_cmu0;
    q.sqrt2 = sqrt(2.0);
    q.cmu_fac1 = pow(cmu0, (-p / n)); q.cmu_fac2 = pow(cmu0, (3.0 + p / n)); q.cmu_fac3 = 1.0 / pow(cmu0, 2.0);
    q.fac2 = pow(cmu0, p) * n * pow(q.vonKar, n); q.fac3 = pow(cmu0, p) * n; q.fac4 = pow(cmu0, p);
    q.fac5 = pow(0.56, 0.5 * n) * pow(cmu0, p); q.fac6 = 8.0 / pow(cmu0, 6.0);
    q.gls_exp1 = 1.0 / n; q.tke_exp1 = m / n; q.tke_exp2 = 0.5 + m / n; q.tke_exp4 = m + 0.5 * n;
    if (G.stability == RH_GLS_CANUTO_A || G.stability == RH_GLS_CANUTO_B) {
        q.s0 = 3.0 / 2.0 * L1 * L5 * L5;
        q.s1 = -L4 * (L6 + L7) + 2.0 * L4 * L5 * (L1 - 1.0 / 3.0 * L2 - L3) + 3.0 / 2.0 * L1 * L5 * L8;
        q.s2 = -3.0 / 8.0 * L1 * (L6 * L6 - L7 * L7);
        q.s4 = 2.0 * L5; q.s5 = 2.0 * L4;
        q.s6 = 2.0 / 3.0 * L5 * (3.0 * L3 * L3 - L2 * L2) - 1.0 / 2.0 * L5 * L1 * (3.0 * L3 - L2) + 3.0 / 4.0 * L1 * (L6 - L7);
        q.b0 = 3.0 * L5 * L5; q.b1 = L5 * (7.0 * L4 + 3.0 * L8);
        q.b2 = L5 * L5 * (3.0 * L3 * L3 - L2 * L2) - 3.0 / 4.0 * (L6 * L6 - L7 * L7);
        q.b3 = L4 * (4.0 * L4 + 3.0 * L8);
        q.b5 = 1.0 / 4.0 * (L2 * L2 - 3.0 * L3 * L3) * (L6 * L6 - L7 * L7);
        q.b4 = L4 * (L2 * L6 - 3.0 * L3 * L7 - L5 * (L2 * L2 - L3 * L3)) + L5 * L8 * (3.0 * L3 * L3 - L2 * L2);
    } else {
        const double A1 = 0.92, A2 = 0.74, B1 = 16.6, B2 = 10.1, C1 = 0.08;
        q.Sm2 = 9.0 * A1 * A2; q.Sm3 = A1 * (1.0 - 3.0 * C1 - 6.0 * A1 / B1); q.Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2;
        q.Sh1 = A2 * (1.0 - 6.0 * A1 / B1); q.Sh2 = 3.0 * A2 * (6.0 * A1 + B2);
    }
    const double Zos_min = std::max(G.Zos, 0.0001); q.Zob_min = std::max(G.Zob, 0.0001); q.Zos_eff = Zos_min;
    q.Gadv = 1.0 / 3.0; q.eps = 1.0e-10;
    q.cmu0_p = pow(cmu0, p); q.surf_len_n = pow(q.L_sft * q.Zos_eff, n); q.bot_len_n = pow(q.vonKar * q.Zob_min, n); q.Lsft_n = pow(q.L_sft, (n));
    if (!c->gls_libm_pow) {
        q.k_m = gpow_class(m); q.k_n = gpow_class(n); q.k_mn = gpow_class(-n); q.k_nm1 = gpow_class(n - 1.0); q.k_e1 = gpow_class(q.gls_exp1);
        q.k_me1 = gpow_class(-q.gls_exp1); q.k_mt1 = gpow_class(-q.tke_exp1); q.k_t2 = gpow_class(q.tke_exp2); q.k_t4 = gpow_class(q.tke_exp4);
    }
    return q;
}

// ---------------------------------------------------------------------------------------------------------------- predictor
// third-order flux of q(.,.,k,n) through the x-face i (:74-98) / the y-face j (:101-125); tf = transport through the face
__device__ __forceinline__ double pre_flux_x(const A4 &q, int n, const A4 &msku, int i, int j, int k, double tf, const GlsC &g)
{
    double gm = (q(i - 1, j, k, n) - q(i - 2, j, k, n)) * msku(i - 1, j, 0);
    double gp = (q(i + 1, j, k, n) - q(i, j, k, n)) * msku(i + 1, j, 0);
    if (i == g.dlo0 - 1 && !g.perx) gm = q(i, j, k, n) - q(i - 1, j, k, n);
    else if (i == g.dhi0 + 1 && !g.perx) gp = q(i, j, k, n) - q(i - 1, j, k, n);
    const double cff = 1.0 / 6.0;
    return tf * 0.5 * (q(i - 1, j, k, n) + q(i, j, k, n) - cff * (gp - gm));
}
__device__ __forceinline__ double pre_flux_y(const A4 &q, int n, const A4 &mskv, int i, int j, int k, double tf, const GlsC &g)
{
    double gm = (q(i, j - 1, k, n) - q(i, j - 2, k, n)) * mskv(i, j - 1, 0);
    double gp = (q(i, j + 1, k, n) - q(i, j, k, n)) * mskv(i, j + 1, 0);
    if (j == g.dlo1 - 1 && !g.pery) gm = q(i, j, k, n) - q(i, j - 1, k, n);
    else if (j == g.dhi1 + 1 && !g.pery) gp = q(i, j, k, n) - q(i, j - 1, k, n);
    const double cff = 1.0 / 6.0;
    return tf * 0.5 * (q(i, j - 1, k, n) + q(i, j, k, n) - cff * (gp - gm));
}
// vertical flux of component n through rho-level k (k = 0..N), given cf = the transport there (:161-193 and :620-640)
template <class Q> __device__ __forceinline__ double vert_flux(const Q &q, int n, int i, int j, int k, int N, double cf)
{
    if (k == 0) { const double a = 1.0 / 3.0, b = 5.0 / 6.0, c = 1.0 / 6.0; return cf * (a * q(i, j, 0, n) + b * q(i, j, 1, n) - c * q(i, j, 2, n)); }
    if (k == N) { const double a = 1.0 / 3.0, b = 5.0 / 6.0, c = 1.0 / 6.0; return cf * (a * q(i, j, k + 1, n) + b * q(i, j, k, n) - c * q(i, j, k - 1, n)); }
    const double a = 7.0 / 12.0, b = 1.0 / 12.0;
    return cf * (a * (q(i, j, k, n) + q(i, j, k + 1, n)) - b * (q(i, j, k - 1, n) + q(i, j, k + 2, n)));
}

// one thread per column of the valid box, marching k = 1..N (REMORA_gls.cpp:74-209)
__global__ void __launch_bounds__(RH_BX * RH_BY)
k_gls_prestep(B3 bx, A4 tke, A4 gls, A4 W, A4 Huon, A4 Hvom, A4 Hz, A4 pm, A4 pn, A4 msku, A4 mskv,
              int nstp, int nnew, int indx, double cff1, double cff2, double cff3, GlsC g)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    const int N = g.N;
    const double cff4 = cff3 * pm(i, j, 0) * pn(i, j, 0);
    double cf_m = 0.5 * (W(i, j, 1) + W(i, j, 0));                        // CF, FC, FCL at rho-level k-1
    double fc_m = vert_flux(tke, nstp, i, j, 0, N, cf_m), fl_m = vert_flux(gls, nstp, i, j, 0, N, cf_m);
    for (int k = 1; k <= N; k++) {
        const double xf0 = 0.5 * (Huon(i, j, k) + Huon(i, j, k - 1)), xf1 = 0.5 * (Huon(i + 1, j, k) + Huon(i + 1, j, k - 1));
        const double ef0 = 0.5 * (Hvom(i, j, k) + Hvom(i, j, k - 1)), ef1 = 0.5 * (Hvom(i, j + 1, k) + Hvom(i, j + 1, k - 1));
        const double fx0 = pre_flux_x(tke, nstp, msku, i, j, k, xf0, g), fx1 = pre_flux_x(tke, nstp, msku, i + 1, j, k, xf1, g);
        const double fe0 = pre_flux_y(tke, nstp, mskv, i, j, k, ef0, g), fe1 = pre_flux_y(tke, nstp, mskv, i, j + 1, k, ef1, g);
        const double lx0 = pre_flux_x(gls, nstp, msku, i, j, k, xf0, g), lx1 = pre_flux_x(gls, nstp, msku, i + 1, j, k, xf1, g);
        const double le0 = pre_flux_y(gls, nstp, mskv, i, j, k, ef0, g), le1 = pre_flux_y(gls, nstp, mskv, i, j + 1, k, ef1, g);
        const double cff = 0.5 * (Hz(i, j, k) + Hz(i, j, k - 1));
        double hzh = cff - cff4 * (xf1 - xf0 + ef1 - ef0);
        const double tn = tke(i, j, k, nstp), gn = gls(i, j, k, nstp);
        double t2 = cff * (cff1 * tn + cff2 * tke(i, j, k, indx)) - cff4 * (fx1 - fx0 + fe1 - fe0);
        double g2 = cff * (cff1 * gn + cff2 * gls(i, j, k, indx)) - cff4 * (lx1 - lx0 + le1 - le0);
        const double cf_k = 0.5 * (W(i, j, k + 1) + W(i, j, k));
        const double fc_k = vert_flux(tke, nstp, i, j, k, N, cf_k), fl_k = vert_flux(gls, nstp, i, j, k, N, cf_k);
        hzh = hzh - cff4 * (cf_k - cf_m);
        const double inv = 1.0 / hzh;
        t2 = inv * (t2 - cff4 * (fc_k - fc_m));
        g2 = inv * (g2 - cff4 * (fl_k - fl_m));
        tke(i, j, k, 2) = t2; gls(i, j, k, 2) = g2;
        tke(i, j, k, nnew) = cff * tn; gls(i, j, k, nnew) = cff * gn;
        cf_m = cf_k; fc_m = fc_k; fl_m = fl_k;
    }
}

// ---- the predictor with the horizontal stencils in LDS (same block shape and flux hand-over as k_gls_hadv_tiled) ----------------
template <class Q> __device__ __forceinline__ double pre_flux_x_t(const Q &q, const A4 &msku, int i, int j, int k, double tf, const GlsC &g)
{
    double gm = (q(i - 1, j, k) - q(i - 2, j, k)) * msku(i - 1, j, 0);
    double gp = (q(i + 1, j, k) - q(i, j, k)) * msku(i + 1, j, 0);
    if (i == g.dlo0 - 1 && !g.perx) gm = q(i, j, k) - q(i - 1, j, k);
    else if (i == g.dhi0 + 1 && !g.perx) gp = q(i, j, k) - q(i - 1, j, k);
    const double cff = 1.0 / 6.0;
    return tf * 0.5 * (q(i - 1, j, k) + q(i, j, k) - cff * (gp - gm));
}
template <class Q> __device__ __forceinline__ double pre_flux_y_t(const Q &q, const A4 &mskv, int i, int j, int k, double tf, const GlsC &g)
{
    double gm = (q(i, j - 1, k) - q(i, j - 2, k)) * mskv(i, j - 1, 0);
    double gp = (q(i, j + 1, k) - q(i, j, k)) * mskv(i, j + 1, 0);
    if (j == g.dlo1 - 1 && !g.pery) gm = q(i, j, k) - q(i, j - 1, k);
    else if (j == g.dhi1 + 1 && !g.pery) gp = q(i, j, k) - q(i, j - 1, k);
    const double cff = 1.0 / 6.0;
    return tf * 0.5 * (q(i, j - 1, k) + q(i, j, k) - cff * (gp - gm));
}
// vertical flux through rho-level k from the column window (qm, q0, q1, q2) = q(k-1), q(k), q(k+1), q(k+2)
__device__ __forceinline__ double vert_flux_w(int k, int N, double cf, double qm, double q0, double q1, double q2)
{
    if (k == N) { const double a = 1.0 / 3.0, b = 5.0 / 6.0, c = 1.0 / 6.0; return cf * (a * q1 + b * q0 - c * qm); }
    const double a = 7.0 / 12.0, b = 1.0 / 12.0;
    return cf * (a * (q0 + q1) - b * (qm + q2));
}
__global__ void __launch_bounds__(RT_BX * RT_BY)
k_gls_prestep_tiled(B3 bx, A4 tks, A4 gss, A4 tki, A4 gsi, A4 tk2, A4 gs2, A4 tkn, A4 gsn, A4 W, A4 Huon, A4 Hvom, A4 Hz, A4 pm, A4 pn,
                    A4 msku, A4 mskv, double cff1, double cff2, double cff3, GlsC g)
{   // tks / gss: time level nstp; tki / gsi: level indx; tk2 / gs2: predictor (component 2); tkn / gsn: level nnew
    typedef Tile<2, 2> T;
    constexpr int NT = RT_BX * RT_BY;
    __shared__ double lds[4 * T::SIZE + 6 * NT];
    double *fl = lds + 4 * T::SIZE;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * RT_BX + tx;
    const RhTile rt = rh_tile_patch<1, 4>();
    const int I0 = bx.lo0 + rt.x * GH_SX, J0 = bx.lo1 + rt.y * GH_SY, i = I0 + tx, j = J0 + ty;
    const bool owner = tx < GH_SX && ty < GH_SY && i <= bx.hi0 && j <= bx.hi1;
    const bool fx_on = i <= bx.hi0 + 1 && j <= bx.hi1, fy_on = i <= bx.hi0 && j <= bx.hi1 + 1;
    const HaloCell hc = halo_cell<2, 2>(tks, I0, J0, tx, ty);
    const int N = g.N;
    const double cff4 = owner ? cff3 * pm(i, j, 0) * pn(i, j, 0) : 0.0;
    // column windows of the own point: levels k-1 .. k+2 (k = 1 first), the level after them in flight
    double tm = ld_in(tks, i, j, 0), t0 = ld_in(tks, i, j, 1), t1 = ld_in(tks, i, j, 2), t2 = N >= 2 ? ld_in(tks, i, j, 3) : 0.0;
    double gm = ld_in(gss, i, j, 0), g0 = ld_in(gss, i, j, 1), g1 = ld_in(gss, i, j, 2), g2 = N >= 2 ? ld_in(gss, i, j, 3) : 0.0;
    double t_h = halo_ld(tks, hc, 1), g_h = halo_ld(gss, hc, 1);
    double hu_m = fx_on ? Huon(i, j, 0) : 0.0, hv_m = fy_on ? Hvom(i, j, 0) : 0.0, hu = fx_on ? Huon(i, j, 1) : 0.0, hv = fy_on ? Hvom(i, j, 1) : 0.0;
    double hz_m = 0.0, hz = 0.0, w_k = 0.0, w_p = 0.0, ti = 0.0, gi = 0.0, cf_m = 0.0, fc_m = 0.0, fl_m = 0.0;
    if (owner) {
        hz_m = Hz(i, j, 0); hz = Hz(i, j, 1); w_k = W(i, j, 1); w_p = W(i, j, 2); ti = tki(i, j, 1); gi = gsi(i, j, 1);
        cf_m = 0.5 * (w_k + W(i, j, 0));                                        // rho-level 0 (:165-174)
        const double a = 1.0 / 3.0, b = 5.0 / 6.0, c = 1.0 / 6.0;
        fc_m = cf_m * (a * tm + b * t0 - c * t1); fl_m = cf_m * (a * gm + b * g0 - c * g1);
    }
    for (int k = 1; k <= N; k++) {
        double *bt = lds + (k & 1) * 2 * T::SIZE, *bg = bt + T::SIZE;
        tile_put<2, 2>(bt, tx, ty, t0, hc, t_h); tile_put<2, 2>(bg, tx, ty, g0, hc, g_h);
        const double hu_c = hu, hv_c = hv, hz_c = hz, ti_c = ti, gi_c = gi, w_c = w_k, w_n = w_p;
        double t3 = 0.0, g3 = 0.0;
        if (k < N) {                                                             // next level's operands
            if (k + 3 <= N + 1) { t3 = ld_in(tks, i, j, k + 3); g3 = ld_in(gss, i, j, k + 3); }
            t_h = halo_ld(tks, hc, k + 1); g_h = halo_ld(gss, hc, k + 1);
            if (fx_on) hu = Huon(i, j, k + 1);
            if (fy_on) hv = Hvom(i, j, k + 1);
            if (owner) { hz = Hz(i, j, k + 1); w_p = W(i, j, k + 2); ti = tki(i, j, k + 1); gi = gsi(i, j, k + 1); }
        }
        __syncthreads();
        const TView<2, 2> vt{bt, I0, J0}, vg{bg, I0, J0};
        double xf = 0.0, ef = 0.0, fx = 0.0, fe = 0.0, lx = 0.0, le = 0.0;
        if (fx_on) { xf = 0.5 * (hu_c + hu_m); fx = pre_flux_x_t(vt, msku, i, j, k, xf, g); lx = pre_flux_x_t(vg, msku, i, j, k, xf, g); }
        if (fy_on) { ef = 0.5 * (hv_c + hv_m); fe = pre_flux_y_t(vt, mskv, i, j, k, ef, g); le = pre_flux_y_t(vg, mskv, i, j, k, ef, g); }
        fl[tid] = xf; fl[NT + tid] = fx; fl[2 * NT + tid] = lx; fl[3 * NT + tid] = ef; fl[4 * NT + tid] = fe; fl[5 * NT + tid] = le;
        __syncthreads();
        if (owner) {
            const double xf1 = fl[tid + 1], fx1 = fl[NT + tid + 1], lx1 = fl[2 * NT + tid + 1];
            const double ef1 = fl[3 * NT + tid + RT_BX], fe1 = fl[4 * NT + tid + RT_BX], le1 = fl[5 * NT + tid + RT_BX];
            const double cff = 0.5 * (hz_c + hz_m);
            double hzh = cff - cff4 * (xf1 - xf + ef1 - ef);
            double p2 = cff * (cff1 * t0 + cff2 * ti_c) - cff4 * (fx1 - fx + fe1 - fe);
            double q2 = cff * (cff1 * g0 + cff2 * gi_c) - cff4 * (lx1 - lx + le1 - le);
            const double cf_k = 0.5 * (w_n + w_c);
            const double fc_k = vert_flux_w(k, N, cf_k, tm, t0, t1, t2), fl_k = vert_flux_w(k, N, cf_k, gm, g0, g1, g2);
            hzh = hzh - cff4 * (cf_k - cf_m);
            const double inv = 1.0 / hzh;
            p2 = inv * (p2 - cff4 * (fc_k - fc_m));
            q2 = inv * (q2 - cff4 * (fl_k - fl_m));
            tk2(i, j, k) = p2; gs2(i, j, k) = q2;
            tkn(i, j, k) = cff * t0; gsn(i, j, k) = cff * g0;
            cf_m = cf_k; fc_m = fc_k; fl_m = fl_k;
        }
        hu_m = hu_c; hv_m = hv_c; hz_m = hz_c; w_k = w_n;
        tm = t0; t0 = t1; t1 = t2; t2 = t3; gm = g0; g0 = g1; g1 = g2; g2 = g3;
    }
}

extern "C" int rh_gls_prestep(rh_ctx *c, rh_state *s, rh_array4 W, int nstp, int nnew, int iic, int ntfirst)
{
    TScope t(c, "gls_prestep");
    if (c->prm.vmix_mode != RH_VMIX_GLS) return rh_fail("rh_gls_prestep: the context was not created with vmix_mode = RH_VMIX_GLS");
    if (nstp < 0 || nstp > 1 || nnew != 1 - nstp) return rh_fail("rh_gls_prestep: nstp/nnew must be the two time levels 0/1");
    const rh_box bx = c->bx, wbx = bx_nodal(bx, 2);
    const rh_box g2 = bx_grow(wbx, 2, 2, 0);
    RH_NEED(s->tke, g2); RH_NEED(s->gls, g2); RH_NEED(W, wbx);
    if (s->tke.ncomp < 3 || s->gls.ncomp < 3) return rh_fail("rh_gls_prestep: tke / gls need 3 components");
    RH_NEED(s->Huon, bx_nodal(bx, 0)); RH_NEED(s->Hvom, bx_nodal(bx, 1)); RH_NEED(s->Hz, bx);
    RH_NEED(s->pm, bx_slab(bx, 2, 0)); RH_NEED(s->pn, bx_slab(bx, 2, 0));
    RH_NEED(s->msku, bx_slab(bx_grow(bx_nodal(bx, 0), 1, 0, 0), 2, 0)); RH_NEED(s->mskv, bx_slab(bx_grow(bx_nodal(bx, 1), 0, 1, 0), 2, 0));
    const double dt = c->prm.dt, gamma = 1.0 / 6.0;
    double cff1, cff2, cff3; int indx;
    if (iic == ntfirst) { cff1 = 1.0; cff2 = 0.0; cff3 = 0.5 * dt; indx = nstp; }
    else { cff1 = 0.5 + gamma; cff2 = 0.5 - gamma; cff3 = (1.0 - gamma) * dt; indx = 1 - nstp; }
    const GlsC g = gls_constants(c);
    if (c->unfused || c->N < 2)
        k_gls_prestep<<<grid2(bx), rh_block(), 0, c->stream>>>(mkB(bx), mkA(s->tke), mkA(s->gls), mkA(W), mkA(s->Huon), mkA(s->Hvom), mkA(s->Hz),
            mkA(s->pm), mkA(s->pn), mkA(s->msku), mkA(s->mskv), nstp, nnew, indx, cff1, cff2, cff3, g);
    else {
        const dim3 grid((bx.hi[0] - bx.lo[0] + GH_SX) / GH_SX, (bx.hi[1] - bx.lo[1] + GH_SY) / GH_SY, 1);
        k_gls_prestep_tiled<<<grid, rt_block(), 0, c->stream>>>(mkB(bx), mkA(s->tke, nstp), mkA(s->gls, nstp), mkA(s->tke, indx), mkA(s->gls, indx),
            mkA(s->tke, 2), mkA(s->gls, 2), mkA(s->tke, nnew), mkA(s->gls, nnew), mkA(W), mkA(s->Huon), mkA(s->Hvom), mkA(s->Hz),
            mkA(s->pm), mkA(s->pn), mkA(s->msku), mkA(s->mskv), cff1, cff2, cff3, g);
    }
    RH_LAUNCH_CHECK();
    BcScope bcs(c);                                                               // :212-215
    for (int ic = 0; ic < 3; ic++) {
        RH_TRY(halo_fillpatch(c, mfv(c, s->tke, 0, 0, 1), s->mskr, RH_BCV_ZVEL, ic, 1));
        RH_TRY(halo_fillpatch(c, mfv(c, s->gls, 0, 0, 1), s->mskr, RH_BCV_ZVEL, ic, 1));
    }
    return bcs.end();
}

// ---------------------------------------------------------------------------------------------------------------- corrector
// vertical shear by a spline through the cell-centred velocities (:406-429): one thread per column of the grow-1 box
__global__ void __launch_bounds__(RH_BX * RH_BY)
k_gls_shear(B3 gb, A4 Hz, A4 u, A4 v, A4 wCF, A4 wdU, A4 wdV, A4 S2C, int N)
{
    RH_IJ(gb); if (!inb2(gb, i, j)) return;
    double cf = 0.0, du = 0.0, dv = 0.0;
    double hz_m = Hz(i, j, 0), u0_m = u(i, j, 0), u1_m = u(i + 1, j, 0), v0_m = v(i, j, 0), v1_m = v(i, j + 1, 0);
    for (int k = 1; k <= N; k++) {
        const double hz = Hz(i, j, k), u0 = u(i, j, k), u1 = u(i + 1, j, k), v0 = v(i, j, k), v1 = v(i, j + 1, k);
        const double cff = 1.0 / (2.0 * hz + hz_m * (2.0 - cf));
        cf = cff * hz;
        du = cff * (3.0 * (u0 - u0_m + u1 - u1_m) - hz_m * du);
        dv = cff * (3.0 * (v0 - v0_m + v1 - v1_m) - hz_m * dv);
        wCF(i, j, k) = cf; wdU(i, j, k) = du; wdV(i, j, k) = dv;
        hz_m = hz; u0_m = u0; u1_m = u1; v0_m = v0; v1_m = v1;
    }
    du = 0.0; dv = 0.0;                                                           // dU, dV at N+1
    for (int k = N; k >= 1; k--) {
        const double cfk = wCF(i, j, k);
        du = wdU(i, j, k) - cfk * du;
        dv = wdV(i, j, k) - cfk * dv;
        S2C(i, j, k) = du * du + dv * dv;
    }
    S2C(i, j, 0) = 0.0;
}

// buoyancy frequency and shear on w-points: rho -> psi -> rho averages (:527-537)
__global__ void k_gls_avg(B3 b, A4 bvf, A4 S2C, A4 B2, A4 S2)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const int k = b.lo2 + RH_KZ;
    auto psi = [&](const A4 &q, int ii, int jj) { return 0.25 * (q(ii, jj, k) + q(ii + 1, jj, k) + q(ii, jj + 1, k) + q(ii + 1, jj + 1, k)); };
    B2(i, j, k) = 0.25 * (psi(bvf, i, j) + psi(bvf, i - 1, j) + psi(bvf, i, j - 1) + psi(bvf, i - 1, j - 1));
    S2(i, j, k) = 0.25 * (psi(S2C, i, j) + psi(S2C, i - 1, j) + psi(S2C, i, j - 1) + psi(S2C, i - 1, j - 1));
}

// curvature of the predictor (component 2) at cell i along x / cell j along y (:540-563, :575-593)
__device__ __forceinline__ double curv_x(const A4 &q, const A4 &msku, int i, int j, int k, const GlsC &g)
{
    double gr, gr1;
    if (i == g.dlo0 - 1 && !g.perx) { gr1 = q(i + 1, j, k, 2) - q(i, j, k, 2); gr = gr1; }
    else if (i == g.dhi0 + 1 && !g.perx) { gr = q(i, j, k, 2) - q(i - 1, j, k, 2); gr1 = gr; }
    else { gr = (q(i, j, k, 2) - q(i - 1, j, k, 2)) * msku(i, j, 0); gr1 = (q(i + 1, j, k, 2) - q(i, j, k, 2)) * msku(i + 1, j, 0); }
    return gr1 - gr;
}
__device__ __forceinline__ double curv_y(const A4 &q, const A4 &mskv, int i, int j, int k, const GlsC &g)
{
    double gr = (q(i, j, k, 2) - q(i, j - 1, k, 2)) * mskv(i, j, 0);
    double gr1 = (q(i, j + 1, k, 2) - q(i, j, k, 2)) * mskv(i, j + 1, 0);
    if (j == g.dlo1 - 1 && !g.pery) gr = gr1;
    else if (j == g.dhi1 + 1 && !g.pery) gr1 = gr;
    return gr1 - gr;
}
// upstream-biased third-order horizontal advection of the predictor into the new time level (:564-616)
__global__ void k_gls_hadv(B3 b, A4 tke, A4 gls, A4 Huon, A4 Hvom, A4 pm, A4 pn, A4 msku, A4 mskv, int nnew, double dt, GlsC g)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const int k = b.lo2 + RH_KZ;
    double fxk[2], fxp[2], fek[2], fep[2];
    #pragma unroll
    for (int d = 0; d < 2; d++) {
        const int ii = i + d, jj = j + d;
        const double cx = 0.5 * (Huon(ii, j, k) + Huon(ii, j, k - 1));
        const int iu = (cx > 0.0) ? ii - 1 : ii;
        fxk[d] = cx * 0.5 * (tke(ii - 1, j, k, 2) + tke(ii, j, k, 2) - g.Gadv * curv_x(tke, msku, iu, j, k, g));
        fxp[d] = cx * 0.5 * (gls(ii - 1, j, k, 2) + gls(ii, j, k, 2) - g.Gadv * curv_x(gls, msku, iu, j, k, g));
        const double cy = 0.5 * (Hvom(i, jj, k) + Hvom(i, jj, k - 1));
        const int ju = (cy > 0.0) ? jj - 1 : jj;
        fek[d] = cy * 0.5 * (tke(i, jj - 1, k, 2) + tke(i, jj, k, 2) - g.Gadv * curv_y(tke, mskv, i, ju, k, g));
        fep[d] = cy * 0.5 * (gls(i, jj - 1, k, 2) + gls(i, jj, k, 2) - g.Gadv * curv_y(gls, mskv, i, ju, k, g));
    }
    const double cff = dt * pm(i, j, 0) * pn(i, j, 0);
    double t = tke(i, j, k, nnew) - cff * (fxk[1] - fxk[0] + fek[1] - fek[0]);
    tke(i, j, k, nnew) = dmax_(t, g.Kmin);
    double p = gls(i, j, k, nnew) - cff * (fxp[1] - fxp[0] + fep[1] - fep[0]);
    gls(i, j, k, nnew) = dmax_(p, g.Pmin);
}

// The same pass with the predictor staged in LDS: one 64 x 8 block marches k over a 63 x 7 patch of columns (blocks overlap by one
// column and one row, so the flux through a patch's last faces is computed by the threads that sit on them and no flux is formed
// twice from global memory).  Per level: tke and gls (component 2) go into halo-2 tiles, double-buffered on the parity of k, with
// the next level's values already in flight; every thread forms the x-flux through its west face and the y-flux through its south
// face from the tiles and leaves them in LDS for its east / north neighbour.  Huon / Hvom are read once per level (the level below
// is carried).  Same expressions as k_gls_hadv: bit-identical.
template <class Q> __device__ __forceinline__ double curv_x_t(const Q &q, const A4 &msku, int i, int j, int k, const GlsC &g)
{
    double gr, gr1;
    if (i == g.dlo0 - 1 && !g.perx) { gr1 = q(i + 1, j, k) - q(i, j, k); gr = gr1; }
    else if (i == g.dhi0 + 1 && !g.perx) { gr = q(i, j, k) - q(i - 1, j, k); gr1 = gr; }
    else { gr = (q(i, j, k) - q(i - 1, j, k)) * msku(i, j, 0); gr1 = (q(i + 1, j, k) - q(i, j, k)) * msku(i + 1, j, 0); }
    return gr1 - gr;
}
template <class Q> __device__ __forceinline__ double curv_y_t(const Q &q, const A4 &mskv, int i, int j, int k, const GlsC &g)
{
    double gr = (q(i, j, k) - q(i, j - 1, k)) * mskv(i, j, 0);
    double gr1 = (q(i, j + 1, k) - q(i, j, k)) * mskv(i, j + 1, 0);
    if (j == g.dlo1 - 1 && !g.pery) gr = gr1;
    else if (j == g.dhi1 + 1 && !g.pery) gr1 = gr;
    return gr1 - gr;
}
__global__ void __launch_bounds__(RT_BX * RT_BY)
k_gls_hadv_tiled(B3 b, A4 tke2, A4 gls2, A4 tkn, A4 gln, A4 Huon, A4 Hvom, A4 pm, A4 pn, A4 msku, A4 mskv, double dt, GlsC g)
{
    typedef Tile<2, 2> T;
    constexpr int NT = RT_BX * RT_BY;
    __shared__ double lds[4 * T::SIZE + 4 * NT];
    double *fl = lds + 4 * T::SIZE;
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * RT_BX + tx;
    const RhTile rt = rh_tile_patch<1, 4>();
    const int I0 = b.lo0 + rt.x * GH_SX, J0 = b.lo1 + rt.y * GH_SY, i = I0 + tx, j = J0 + ty;
    const bool owner = tx < GH_SX && ty < GH_SY && i <= b.hi0 && j <= b.hi1;
    const bool fx_on = i <= b.hi0 + 1 && j <= b.hi1, fy_on = i <= b.hi0 && j <= b.hi1 + 1;
    const HaloCell hc = halo_cell<2, 2>(tke2, I0, J0, tx, ty);
    const int k0 = b.lo2, k1 = b.hi2;
    double t_own = ld_in(tke2, i, j, k0), g_own = ld_in(gls2, i, j, k0), t_h = halo_ld(tke2, hc, k0), g_h = halo_ld(gls2, hc, k0);
    double hu_m = fx_on ? Huon(i, j, k0 - 1) : 0.0, hv_m = fy_on ? Hvom(i, j, k0 - 1) : 0.0;
    double hu = fx_on ? Huon(i, j, k0) : 0.0, hv = fy_on ? Hvom(i, j, k0) : 0.0;
    double tn = owner ? tkn(i, j, k0) : 0.0, gn = owner ? gln(i, j, k0) : 0.0;
    const double cff = owner ? dt * pm(i, j, 0) * pn(i, j, 0) : 0.0;
    for (int k = k0; k <= k1; k++) {
        double *bt = lds + (k & 1) * 2 * T::SIZE, *bg = bt + T::SIZE;
        tile_put<2, 2>(bt, tx, ty, t_own, hc, t_h); tile_put<2, 2>(bg, tx, ty, g_own, hc, g_h);
        const double hu_c = hu, hv_c = hv, tn_c = tn, gn_c = gn;
        if (k < k1) {                                                            // next level: in flight during this one
            t_own = ld_in(tke2, i, j, k + 1); g_own = ld_in(gls2, i, j, k + 1); t_h = halo_ld(tke2, hc, k + 1); g_h = halo_ld(gls2, hc, k + 1);
            if (fx_on) hu = Huon(i, j, k + 1);
            if (fy_on) hv = Hvom(i, j, k + 1);
            if (owner) { tn = tkn(i, j, k + 1); gn = gln(i, j, k + 1); }
        }
        __syncthreads();
        const TView<2, 2> vt{bt, I0, J0}, vg{bg, I0, J0};
        double fxk = 0.0, fxp = 0.0, fek = 0.0, fep = 0.0;
        if (fx_on) {
            const double cx = 0.5 * (hu_c + hu_m);
            const int iu = (cx > 0.0) ? i - 1 : i;
            fxk = cx * 0.5 * (vt(i - 1, j, k) + vt(i, j, k) - g.Gadv * curv_x_t(vt, msku, iu, j, k, g));
            fxp = cx * 0.5 * (vg(i - 1, j, k) + vg(i, j, k) - g.Gadv * curv_x_t(vg, msku, iu, j, k, g));
        }
        if (fy_on) {
            const double cy = 0.5 * (hv_c + hv_m);
            const int ju = (cy > 0.0) ? j - 1 : j;
            fek = cy * 0.5 * (vt(i, j - 1, k) + vt(i, j, k) - g.Gadv * curv_y_t(vt, mskv, i, ju, k, g));
            fep = cy * 0.5 * (vg(i, j - 1, k) + vg(i, j, k) - g.Gadv * curv_y_t(vg, mskv, i, ju, k, g));
        }
        fl[tid] = fxk; fl[NT + tid] = fxp; fl[2 * NT + tid] = fek; fl[3 * NT + tid] = fep;
        __syncthreads();
        if (owner) {
            const double t = tn_c - cff * (fl[tid + 1] - fxk + fl[2 * NT + tid + RT_BX] - fek);
            tkn(i, j, k) = dmax_(t, g.Kmin);
            const double p = gn_c - cff * (fl[NT + tid + 1] - fxp + fl[3 * NT + tid + RT_BX] - fep);
            gln(i, j, k) = dmax_(p, g.Pmin);
        }
        hu_m = hu_c; hv_m = hv_c;
    }
}

// One thread per column (:620-902): vertical advection, production / dissipation and the forward elimination for tke top-down;
// tke back-substitution; the same for gls (its surface / bottom fluxes need the finished tke); then, riding the gls
// back-substitution upward, the length-scale limit, stability functions and the new Akv / Akt / Akk / Akp / Lscale.
// wA = elimination coefficients (tke, then gls), wB = the gls diagonal BCP kept between the two sweeps.
__global__ void __launch_bounds__(RH_BX * RH_BY)
k_gls_column(B3 bx, A4 tke_a, A4 gls_a, A4 W_a, A4 Akv_a, A4 Akt_a, A4 Akk_a, A4 Akp_a, A4 Lscale_a, A4 Hz_a, A4 pm, A4 pn, A4 z_w_a,
             A4 sustr, A4 svstr, A4 bustr, A4 bvstr, A4 B2_a, A4 S2_a, A4 wA_a, A4 wB_a, int nstp, int nnew, double dt, GlsC g)
{
    RH_IJ(bx); if (!inb2(bx, i, j)) return;
    // column views: the (i, j) part of every 3-D access of this kernel is this thread's own column (rh_tile.h, column cursors)
    const ColV tke = col_view(tke_a, i, j), gls = col_view(gls_a, i, j), W = col_view(W_a, i, j), Akv = col_view(Akv_a, i, j), Akt = col_view(Akt_a, i, j),
               Akk = col_view(Akk_a, i, j), Akp = col_view(Akp_a, i, j), Lscale = col_view(Lscale_a, i, j), Hz = col_view(Hz_a, i, j), z_w = col_view(z_w_a, i, j),
               B2 = col_view(B2_a, i, j), S2 = col_view(S2_a, i, j), wA = col_view(wA_a, i, j), wB = col_view(wB_a, i, j);
    const int N = g.N;
    const double cffA = dt * pm(i, j, 0) * pn(i, j, 0), cffd = -0.5 * dt;
    // Dirichlet values at the surface and the bottom (:732-745)
    const double su = sustr(i, j, 0) + sustr(i + 1, j, 0), sv = svstr(i, j, 0) + svstr(i, j + 1, 0);
    const double bu = bustr(i, j, 0) + bustr(i + 1, j, 0), bv = bvstr(i, j, 0) + bvstr(i, j + 1, 0);
    const double tke_top = dmax_(g.cmu_fac3 * 0.5 * sqrt(su * su + sv * sv), g.Kmin);
    const double tke_bot = dmax_(g.cmu_fac3 * 0.5 * sqrt(bu * bu + bv * bv), g.Kmin);
    tke(i, j, N + 1, nnew) = tke_top; tke(i, j, 0, nnew) = tke_bot;
    gls(i, j, N + 1, nnew) = dmax_(g.cmu0_p * gpow(tke_top, g.gls_m, g.k_m) * g.surf_len_n, g.Pmin);
    { const double cff_gls = g.fac4 * g.bot_len_n; gls(i, j, 0, nnew) = dmax_(cff_gls * gpow(tke_bot, (g.gls_m), g.k_m), g.Pmin); }

    // ---- sweep A (k = N..1)
    auto diffc = [&](const ColV &Ak, int k) { return (k == 0 || k == N) ? 0.0 : cffd * (Ak(i, j, k) + Ak(i, j, k + 1)) / Hz(i, j, k); };
    double faK = vert_flux(tke, 2, i, j, N, N, 0.5 * (W(i, j, N + 1) + W(i, j, N)));
    double faP = vert_flux(gls, 2, i, j, N, N, 0.5 * (W(i, j, N + 1) + W(i, j, N)));
    double fdK = diffc(Akk, N), fdP = diffc(Akp, N);
    double t_up = 0.0, cf_up = 0.0, cb = 0.0;
    for (int k = N; k >= 1; k--) {
        const double cfm = 0.5 * (W(i, j, k) + W(i, j, k - 1));
        const double faK_m = vert_flux(tke, 2, i, j, k - 1, N, cfm), faP_m = vert_flux(gls, 2, i, j, k - 1, N, cfm);
        const double fdK_m = diffc(Akk, k - 1), fdP_m = diffc(Akp, k - 1);
        double t = tke(i, j, k, nnew) - cffA * (faK - faK_m); t = dmax_(t, g.Kmin);            // :641-648
        double p = gls(i, j, k, nnew) - cffA * (faP - faP_m); p = dmax_(p, g.Pmin);
        // production and dissipation (:665-719)
        const double strat2 = B2(i, j, k), sh2 = S2(i, j, k), akv = Akv(i, j, k), akt = Akt(i, j, k, 0);
        const double tn = tke(i, j, k, nstp), gn = gls(i, j, k, nstp);
        const double c3 = (strat2 > 0.0) ? g.c3m : g.c3p;
        double Kprod = sh2 * (akv - g.Akv_bak) - strat2 * (akt - g.Akt_bak);
        double Pprod = g.c1 * sh2 * (akv - g.Akv_bak) - c3 * strat2 * (akt - g.Akt_bak);
        const double on1 = (Kprod < 0.0) ? 0.0 : 1.0, on2 = (Pprod < 0.0) ? 0.0 : 1.0;
        if (Kprod < 0.0) Kprod = Kprod + strat2 * (akt - g.Akt_bak);
        if (Pprod < 0.0) Pprod = Pprod + c3 * strat2 * (akt - g.Akt_bak);
        const double cff_Hz = 0.5 * (Hz(i, j, k) + Hz(i, j, k - 1));
        t = t + dt * cff_Hz * Kprod;
        p = p + dt * cff_Hz * Pprod * gn / dmax_(tn, g.Kmin);
        double wall_fac = 1.0;
        if (g.Lmy25) {
            wall_fac = 1.0 + g.E2 / (g.vonKar * g.vonKar) *
                       pow(pow(gn, (g.gls_exp1)) * g.cmu_fac1 * pow(tn, -g.tke_exp1) * (1.0 / (z_w(i, j, k) - z_w(i, j, 0))), 2.0) +
                       0.25 / (g.vonKar * g.vonKar) *
                       pow(pow(gn, g.gls_exp1) * g.cmu_fac1 * pow(tn, -g.tke_exp1) * (1.0 / (z_w(i, j, N + 1) - z_w(i, j, k))), 2.0);
        }
        const double diss_g = gpow(gn, (-g.gls_exp1), g.k_me1), diss_t = gpow(tn, g.tke_exp2, g.k_t2);
        const double BCK = cff_Hz * (1.0 + dt * diss_g * g.cmu_fac2 * diss_t + dt * (1.0 - on1) * strat2 * (akt - g.Akt_bak) / tn) - fdK - fdK_m;
        const double BCP = cff_Hz * (1.0 + dt * g.c2 * wall_fac * diss_g * g.cmu_fac2 * diss_t + dt * (1.0 - on2) * c3 * strat2 * (akt - g.Akt_bak) / tn) - fdP - fdP_m;
        // forward elimination for tke (:749-758), tke_fluxt = 0
        if (k == N) { cb = 1.0 / BCK; t = cb * (t + 0.0); }
        else { cb = 1.0 / (BCK - cf_up * fdK); t = cb * (t - fdK * t_up); }
        cf_up = cb * fdK_m;
        wA(i, j, k) = cf_up; wB(i, j, k) = BCP;
        tke(i, j, k, nnew) = t; gls(i, j, k, nnew) = p;
        t_up = t; faK = faK_m; faP = faP_m; fdK = fdK_m; fdP = fdP_m;
    }
    // ---- tke back-substitution (:759-764), tke_fluxb = 0
    double t_lo = t_up - cb * 0.0;
    t_lo = dmax_(t_lo, g.Kmin);
    tke(i, j, 1, nnew) = t_lo;
    const double tke1 = t_lo;
    for (int k = 2; k <= N; k++) {
        double t = tke(i, j, k, nnew) - wA(i, j, k) * t_lo;
        t = dmax_(t, g.Kmin);
        tke(i, j, k, nnew) = t; t_lo = t;
    }
    const double tkeN = t_lo;
    // ---- surface and bottom flux of gls (:767-775)
    double ct = 0.5 * (tke_top + tkeN);
    const double gls_fluxt = dt * g.fac3 * gpow(ct, g.gls_m, g.k_m) * g.Lsft_n * gpow(g.Zos_eff + 0.5 * Hz(i, j, N), g.gls_n - 1.0, g.k_nm1) * 0.5 * (Akp(i, j, N + 1) + Akp(i, j, N));
    ct = 0.5 * (tke_bot + tke1);
    const double gls_fluxb = dt * g.fac2 * gpow(ct, g.gls_m, g.k_m) * gpow(0.5 * Hz(i, j, 0) + g.Zob_min, g.gls_n - 1.0, g.k_nm1) * 0.5 * (Akp(i, j, 0) + Akp(i, j, 1));
    // ---- sweep C: forward elimination for gls (:776-783)
    fdP = diffc(Akp, N);
    double p_up = 0.0, cp = 0.0; cf_up = 0.0;
    for (int k = N; k >= 1; k--) {
        const double fdP_m = diffc(Akp, k - 1), BCP = wB(i, j, k);
        double p = gls(i, j, k, nnew);
        if (k == N) { cp = 1.0 / BCP; p = cp * (p - gls_fluxt); }
        else { cp = 1.0 / (BCP - cf_up * fdP); p = cp * (p - fdP * p_up); }
        cf_up = cp * fdP_m;
        wA(i, j, k) = cf_up;
        gls(i, j, k, nnew) = p;
        p_up = p; fdP = fdP_m;
    }
    // ---- sweep D: gls back-substitution (:784-787) and, level by level, the mixing coefficients (:791-885)
    double p_lo = p_up - cp * gls_fluxb;
    for (int k = 1; k <= N; k++) {
        if (k > 1) p_lo = gls(i, j, k, nnew) - wA(i, j, k) * p_lo;
        const double p_solved = p_lo;                         // what the next level's back-substitution uses
        double t = dmax_(tke(i, j, k, nnew), g.Kmin);
        double p = dmax_(p_solved, g.Pmin);
        const double buoy2 = B2(i, j, k), shear2 = S2(i, j, k);
        if (g.gls_n >= 0.0) p = dmin_(p, g.fac5 * gpow(t, g.tke_exp4, g.k_t4) * gpow(sqrt(dmax_(0.0, buoy2)) + g.eps, -g.gls_n, g.k_mn));
        else p = dmax_(p, g.fac5 * gpow(t, (g.tke_exp4), g.k_t4) * gpow((sqrt(dmax_(0.0, buoy2)) + g.eps), (-g.gls_n), g.k_mn));
        double Ls_lmt;
        const double Ls_unlmt = dmax_(g.eps, gpow(p, (g.gls_exp1), g.k_e1) * g.cmu_fac1 * gpow(t, (-g.tke_exp1), g.k_mt1));
        if (buoy2 > 0.0) Ls_lmt = dmin_(Ls_unlmt, sqrt(0.56 * t / (dmax_(0.0, buoy2) + g.eps)));
        else Ls_lmt = Ls_unlmt;
        p = dmax_(g.cmu0_p * gpow(t, g.gls_m, g.k_m) * gpow(Ls_lmt, g.gls_n, g.k_n), g.Pmin);
        double Sh, Sm;
        double Gh = dmin_(g.Gh0, -buoy2 * Ls_lmt * Ls_lmt / (2.0 * t));
        Gh = dmin_(Gh, Gh - (Gh - g.Ghcri) * (Gh - g.Ghcri) / (Gh + g.Gh0 - 2.0 * g.Ghcri));
        Gh = dmax_(Gh, g.Ghmin);
        if (g.stab == RH_GLS_CANUTO_A || g.stab == RH_GLS_CANUTO_B) {
            double Gm = (g.b0 / g.fac6 - g.b1 * Gh + g.b3 * g.fac6 * (Gh * Gh)) / (g.b2 - g.b4 * g.fac6 * Gh);
            Gm = dmin_(Gm, shear2 * Ls_lmt * Ls_lmt / (2.0 * t));
            const double sc = g.b0 - g.b1 * g.fac6 * Gh + g.b2 * g.fac6 * Gm + g.b3 * g.fac6 * g.fac6 * Gh * Gh -
                              g.b4 * g.fac6 * g.fac6 * Gh * Gm + g.b5 * g.fac6 * g.fac6 * Gm * Gm;
            Sm = (g.s0 - g.s1 * g.fac6 * Gh + g.s2 * g.fac6 * Gm) / sc;
            Sh = (g.s4 - g.s5 * g.fac6 * Gh + g.s6 * g.fac6 * Gm) / sc;
            Sm = dmax_(Sm, 0.0); Sh = dmax_(Sh, 0.0);
            const double c3 = g.cmu0 * g.cmu0 * g.cmu0;
            Sm = Sm * g.sqrt2 / (c3); Sh = Sh * g.sqrt2 / c3;
        } else {
            const double cg = 1.0 - g.Sh2 * Gh;
            Sh = g.Sh1 / cg;
            Sm = (g.Sm3 + Sh * Gh * g.Sm4) / (1.0 - g.Sm2 * Gh);
        }
        const double ql = g.sqrt2 * 0.5 * (Ls_lmt * sqrt(t) + Lscale(i, j, k) * sqrt(tke(i, j, k, nstp)));
        tke(i, j, k, nnew) = t; gls(i, j, k, nnew) = p;
        Akv(i, j, k) = g.Akv_bak + Sm * ql;
        for (int n = 0; n < RH_NCONS; n++) Akt(i, j, k, n) = g.Akt_bak + Sh * ql;
        Akk(i, j, k) = g.Akk_bak + Sm * ql / g.sigk;
        Akp(i, j, k) = g.Akp_bak + Sm * ql * g.ogls_sigp;
        Lscale(i, j, k) = Ls_lmt;
    }
    // ---- surface and bottom values of the mixing coefficients (:886-902)
    const double akv_top = g.Akv_bak + g.L_sft * g.Zos_eff * g.cmu0 * sqrt(tke_top);
    const double akv_bot = g.Akv_bak + g.vonKar * g.Zob_min * g.cmu0 * sqrt(tke_bot);
    Akv(i, j, N + 1) = akv_top; Akv(i, j, 0) = akv_bot;
    Akk(i, j, N + 1) = g.Akk_bak + akv_top / g.sigk; Akk(i, j, 0) = g.Akk_bak + akv_bot / g.sigk;
    Akp(i, j, N + 1) = g.Akp_bak + akv_top * g.ogls_sigp; Akp(i, j, 0) = g.Akp_bak + akv_bot / g.sigp_cb;
    for (int n = 0; n < RH_NCONS; n++) { Akt(i, j, N + 1, n) = g.Akt_bak; Akt(i, j, 0, n) = g.Akt_bak; }
}

extern "C" int rh_gls_corrector(rh_ctx *c, rh_state *s, rh_array4 W, int nstp, int nnew)
{
    TScope t(c, "gls_corrector");
    if (c->prm.vmix_mode != RH_VMIX_GLS) return rh_fail("rh_gls_corrector: the context was not created with vmix_mode = RH_VMIX_GLS");
    if (nstp < 0 || nstp > 1 || nnew != 1 - nstp) return rh_fail("rh_gls_corrector: nstp/nnew must be the two time levels 0/1");
    const int N = c->N; const double dt = c->prm.dt;
    if (N < 2) return rh_fail("rh_gls_corrector: at least 3 vertical levels are needed");
    const rh_box bx = c->bx, wbx = bx_nodal(bx, 2), gbx1 = bx_grow(bx, 1, 1, 0), b2 = bx_slab(bx, 2, 0);
    const rh_box wg2 = bx_grow(wbx, 2, 2, 0);
    rh_box win = wbx; win.lo[2] = 1; win.hi[2] = N;                               // interior w-points k = 1..N
    RH_NEED(s->tke, wg2); RH_NEED(s->gls, wg2); RH_NEED(W, wbx);
    if (s->tke.ncomp < 3 || s->gls.ncomp < 3 || s->Akt.ncomp < RH_NCONS) return rh_fail("rh_gls_corrector: tke / gls need 3 components, Akt %d", RH_NCONS);
    RH_NEED(s->Akv, wbx); RH_NEED(s->Akt, wbx); RH_NEED(s->Akk, wbx); RH_NEED(s->Akp, wbx); RH_NEED(s->Lscale, wbx);
    RH_NEED(s->Hz, gbx1); RH_NEED(s->xvel_old, bx_nodal(gbx1, 0)); RH_NEED(s->yvel_old, bx_nodal(gbx1, 1));
    RH_NEED(s->Huon, bx_nodal(bx, 0)); RH_NEED(s->Hvom, bx_nodal(bx, 1)); RH_NEED(s->z_w, wbx);
    RH_NEED(s->pm, b2); RH_NEED(s->pn, b2);
    RH_NEED(s->sustr, bx_nodal(b2, 0)); RH_NEED(s->bustr, bx_nodal(b2, 0)); RH_NEED(s->svstr, bx_nodal(b2, 1)); RH_NEED(s->bvstr, bx_nodal(b2, 1));
    RH_NEED(s->msku, bx_slab(bx_grow(bx_nodal(bx, 0), 1, 0, 0), 2, 0)); RH_NEED(s->mskv, bx_slab(bx_grow(bx_nodal(bx, 1), 0, 1, 0), 2, 0));
    { rh_box bb = gbx1; bb.lo[2] = 1; bb.hi[2] = N; RH_NEED(s->bvf, bb); }
    RH_NEED(c->gS2C, gbx1); RH_NEED(c->gW1, bx_nodal(gbx1, 2)); RH_NEED(c->gW2, bx_nodal(gbx1, 2)); RH_NEED(c->gW3, bx_nodal(gbx1, 2));
    RH_NEED(c->gB2, bx); RH_NEED(c->gS2, bx);
    const GlsC g = gls_constants(c);
    k_gls_shear<<<grid2(gbx1), rh_block(), 0, c->stream>>>(mkB(gbx1), mkA(s->Hz), mkA(s->xvel_old), mkA(s->yvel_old), mkA(c->gW1), mkA(c->gW2), mkA(c->gW3), mkA(c->gS2C), N);
    RH_LAUNCH_CHECK();
    // :434 -- first-order extrapolation of the shear on every side of the domain, periodic ones included
    RH_TRY(halo_physbc(c, mfv(c, c->gS2C, 0, 0, 0), s->mskr, 0, 1, RH_BCV_FOEXTRAP, 0));
    k_gls_avg<<<grid3(win), rh_block(), 0, c->stream>>>(mkB(win), mkA(s->bvf), mkA(c->gS2C), mkA(c->gB2), mkA(c->gS2));
    RH_LAUNCH_CHECK();
    if (c->unfused)
        k_gls_hadv<<<grid3(win), rh_block(), 0, c->stream>>>(mkB(win), mkA(s->tke), mkA(s->gls), mkA(s->Huon), mkA(s->Hvom), mkA(s->pm), mkA(s->pn),
                                                             mkA(s->msku), mkA(s->mskv), nnew, dt, g);
    else {
        const dim3 grid((bx.hi[0] - bx.lo[0] + GH_SX) / GH_SX, (bx.hi[1] - bx.lo[1] + GH_SY) / GH_SY, 1);
        k_gls_hadv_tiled<<<grid, rt_block(), 0, c->stream>>>(mkB(win), mkA(s->tke, 2), mkA(s->gls, 2), mkA(s->tke, nnew), mkA(s->gls, nnew),
                                                             mkA(s->Huon), mkA(s->Hvom), mkA(s->pm), mkA(s->pn), mkA(s->msku), mkA(s->mskv), dt, g);
    }
    RH_LAUNCH_CHECK();
    k_gls_column<<<grid2(bx), rh_block(), 0, c->stream>>>(mkB(bx), mkA(s->tke), mkA(s->gls), mkA(W), mkA(s->Akv), mkA(s->Akt), mkA(s->Akk), mkA(s->Akp),
        mkA(s->Lscale), mkA(s->Hz), mkA(s->pm), mkA(s->pn), mkA(s->z_w), mkA(s->sustr), mkA(s->svstr), mkA(s->bustr), mkA(s->bvstr),
        mkA(c->gB2), mkA(c->gS2), mkA(c->gW1), mkA(c->gW2), nstp, nnew, dt, g);
    RH_LAUNCH_CHECK();
    BcScope bcs(c);                                                               // :905-914
    for (int ic = 0; ic < 3; ic++) {
        RH_TRY(halo_fillpatch(c, mfv(c, s->tke, 0, 0, 1), s->mskr, RH_BCV_ZVEL, ic, 1));
        RH_TRY(halo_fillpatch(c, mfv(c, s->gls, 0, 0, 1), s->mskr, RH_BCV_ZVEL, ic, 1));
    }
    for (int ic = 0; ic < RH_NCONS; ic++) RH_TRY(halo_fillpatch(c, mfv(c, s->Akt, 0, 0, 1), s->mskr, RH_BCV_ZVEL, ic, 1));
    RH_TRY(halo_fill_boundary(c, mfv(c, s->Akv, 0, 0, 1), 0, 1));
    RH_TRY(halo_fill_boundary(c, mfv(c, s->Akp, 0, 0, 1), 0, 1));
    RH_TRY(halo_fill_boundary(c, mfv(c, s->Akk, 0, 0, 1), 0, 1));
    return bcs.end();
}

// REMORA::init_gls_vmix, Source/Initialization/REMORA_init.cpp:156-190 (+ make_new_level.cpp:517-521)
int launch_gls_init(rh_ctx *c, rh_state *s)
{
    const rh_gls_params &G = c->prm.gls; const int N = c->N;
    RH_TRY(dev_setval(c, s->tke, 0, s->tke.ncomp, G.gls_Kmin)); RH_TRY(dev_setval(c, s->gls, 0, s->gls.ncomp, G.gls_Pmin));
    RH_TRY(dev_setval(c, s->Lscale, 0, 1, 0.0));
    RH_TRY(dev_setval(c, s->Akk, 0, 1, G.Akk_bak)); RH_TRY(dev_setval(c, s->Akp, 0, 1, G.Akp_bak));
    RH_TRY(dev_setval(c, s->Akv, 0, 1, G.Akv_bak)); RH_TRY(dev_setval(c, s->Akt, 0, s->Akt.ncomp, G.Akt_bak));
    for (int k : {0, N + 1}) {
        RH_TRY(dev_setbox(c, s->Akk, 0, bx_slab(bx_of(s->Akk), 2, k), 0.0)); RH_TRY(dev_setbox(c, s->Akp, 0, bx_slab(bx_of(s->Akp), 2, k), 0.0));
        RH_TRY(dev_setbox(c, s->Akv, 0, bx_slab(bx_of(s->Akv), 2, k), 0.0)); RH_TRY(dev_setbox(c, s->Akt, 0, bx_slab(bx_of(s->Akt), 2, k), 0.0));   // Akt: component 0 only
    }
    return 0;
}
