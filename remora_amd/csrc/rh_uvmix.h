// rh_uvmix.h -- harmonic-viscosity stress tensor pieces of REMORA::uv3dmix (REMORA_uv3dmix.cpp:48-78,103-134), shared by the
// 3-D call (rh_setup.hip) and the 2-D call of the barotropic step (rh_step2d.hip).  The k-invariant metric factors are
// gathered once per column; the reference recomputes them at every level.
#pragma once
#include "rh_common.h"

struct MixR { double pm_on_pn, pn_on_pm, a1, a0, b1, b0, on_r, om_r, visc; };
struct MixP { double pmon_p, pnom_p, m125, c1, c0, d1, d0, om_p, on_p, visc; };

// rho point (i,j): REMORA_uv3dmix.cpp:48-59 (UFx) and :103-115 (VFe)
template <class PM, class PN, class VR>
__device__ __forceinline__ MixR mix_r_metrics(const PM &pm, const PN &pn, const VR &visc2_r, int i, int j)
{
    MixR r;
    const double pm0 = pm(i, j, 0), pn0 = pn(i, j, 0);
    r.pm_on_pn = pm0 / pn0; r.pn_on_pm = pn0 / pm0;
    r.a1 = pn0 + pn(i + 1, j, 0); r.a0 = pn(i - 1, j, 0) + pn0;
    r.b1 = pm0 + pm(i, j + 1, 0); r.b0 = pm(i, j - 1, 0) + pm0;
    r.on_r = 1.0 / pn0; r.om_r = 1.0 / pm0;
    r.visc = visc2_r(i, j, 0);
    return r;
}
template <class H, class U, class V>
__device__ __forceinline__ double mix_r_cff(const MixR &r, const H &Hz, const U &u, const V &v, int i, int j, int k)
{
    return 0.5 * Hz(i, j, k) * (r.pm_on_pn * (r.a1 * u(i + 1, j, k) - r.a0 * u(i, j, k)) -
                                r.pn_on_pm * (r.b1 * v(i, j + 1, k) - r.b0 * v(i, j, k)));
}
// psi point (i,j): REMORA_uv3dmix.cpp:61-78 (UFe) and :117-134 (VFx)
template <class PM, class PN, class VP, class MP>
__device__ __forceinline__ MixP mix_p_metrics(const PM &pm, const PN &pn, const VP &visc2_p, const MP &mskp, int i, int j)
{
    MixP p;
    const double spm = pm(i - 1, j - 1, 0) + pm(i - 1, j, 0) + pm(i, j - 1, 0) + pm(i, j, 0);
    const double spn = pn(i - 1, j - 1, 0) + pn(i - 1, j, 0) + pn(i, j - 1, 0) + pn(i, j, 0);
    p.pmon_p = spm / spn; p.pnom_p = spn / spm;
    p.m125 = mskp(i, j, 0) * 0.125;
    p.c1 = pn(i, j - 1, 0) + pn(i, j, 0); p.c0 = pn(i - 1, j - 1, 0) + pn(i - 1, j, 0);
    p.d1 = pm(i - 1, j, 0) + pm(i, j, 0); p.d0 = pm(i - 1, j - 1, 0) + pm(i, j - 1, 0);
    p.om_p = 4.0 / spm; p.on_p = 4.0 / spn;
    p.visc = visc2_p(i, j, 0);
    return p;
}
template <class H, class U, class V>
__device__ __forceinline__ double mix_p_cff(const MixP &p, const H &Hz, const U &u, const V &v, int i, int j, int k)
{
    return p.m125 * (Hz(i - 1, j, k) + Hz(i, j, k) + Hz(i - 1, j - 1, k) + Hz(i, j - 1, k)) *
           (p.pmon_p * (p.c1 * v(i, j, k) - p.c0 * v(i - 1, j, k)) + p.pnom_p * (p.d1 * u(i, j, k) - p.d0 * u(i, j - 1, k)));
}
