// rh_driver.hip -- MultiFab-level drivers of the hot path: REMORA::setup_step (+ prestep), advance_3d_ml / advance_3d, Advance,
// plus the synthetic initial state (what InitData / Exec/<case>/prob.cpp provide in REMORA) and the plotfile variables.
// Host orchestration only: every numeric pass is a kernel in rh_setup.hip / rh_adv3d.hip / rh_step2d.hip / rh_halo.hip.
#include "rh_common.h"

// REMORA::set_smflux -> init_custom_smflux (Exec/Upwelling/prob.cpp:214-255); the other pinned cases keep a time-independent field
static int set_smflux(rh_ctx *c, rh_state *s, double time)
{
    const rh_params &P = c->prm;
    if (P.smflux_mode == RH_SMFLUX_FIXED) return 0;
    if (P.smflux_mode == RH_SMFLUX_NONE) { RH_TRY(dev_setval(c, s->sustr, 0, 1, 0.0)); return dev_setval(c, s->svstr, 0, 1, 0.0); }
    if (P.smflux_mode == RH_SMFLUX_CHANNEL) {                                   // Exec/Channel_Test/prob.cpp:204-222
        const double tdays = time / (24.0 * 60.0 * 60.0), dstart = 0.0, userval = 0.01, fcor = 1.0e-4;
        const double windamp_u = userval * sin(0.9 * fcor * 86400.0 * (tdays - dstart)) / P.rho0;
        RH_TRY(dev_setval(c, s->sustr, 0, 1, windamp_u)); return dev_setval(c, s->svstr, 0, 1, 0.0);
    }
    if (P.smflux_mode != RH_SMFLUX_UPWELLING) return rh_fail("set_smflux: unknown smflux_mode %d", P.smflux_mode);
    const double pi = 3.14159265359, tdays = time / (24.0 * 60.0 * 60.0), dstart = 0.0;
    double windamp;
    if ((tdays - dstart) <= 2.0) windamp = -0.1 * sin(pi * (tdays - dstart) / 4.0) / P.rho0;
    else windamp = -0.1 / P.rho0;
    if (P.periodic[1]) { RH_TRY(dev_setval(c, s->sustr, 0, 1, 0.0)); RH_TRY(dev_setval(c, s->svstr, 0, 1, windamp)); }
    else if (P.periodic[0]) { RH_TRY(dev_setval(c, s->sustr, 0, 1, windamp)); RH_TRY(dev_setval(c, s->svstr, 0, 1, 0.0)); }
    return 0;
}

// REMORA::set_analytical_vmix, Source/REMORA.cpp:565-573
static int set_analytical_vmix(rh_ctx *c, rh_state *s)
{
    // constants that these arrays already hold (written by an earlier step, nothing else writes Akv / Akt without GLS): see rh_ctx::d_vmixrow
    if (c->prm.vmix_mode == RH_VMIX_CONST && !c->vmix_rewrite && c->vmix_akv_done == s->Akv.p && c->vmix_akt_done == s->Akt.p) return 0;
    c->vmix_akv_done = c->vmix_akt_done = nullptr;
    TScope t(c, "vmix");
    RH_TRY(launch_vmix(c, s));
    BcScope bcs(c);
    RH_TRY(halo_fillpatch(c, mfv(c, s->Akv, 0, 0, 1), s->mskr, RH_BCV_ZVEL, 0, 1));
    // the reference loops over NCONS but fills component 0 each time (REMORA.cpp:570-572): once is the same
    RH_TRY(halo_fillpatch(c, mfv(c, s->Akt, 0, 0, 1), s->mskr, RH_BCV_ZVEL, 0, 1));
    RH_TRY(bcs.end());
    if (c->prm.vmix_mode == RH_VMIX_CONST) { c->vmix_akv_done = s->Akv.p; c->vmix_akt_done = s->Akt.p; }
    return 0;
}

__global__ void k_zero_box(B3 b, A4 a)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    a(i, j, b.lo2 + RH_KZ) = 0.0;
}
// a := 0 outside the i-j extent of `inner` (all k): what is left of a setVal(0) when the next kernel overwrites `inner` entirely
static int zero_outside(rh_ctx *c, const rh_array4 &a, const rh_box &inner)
{
    const rh_box all = bx_of(a);
    rh_box r[4] = {all, all, all, all};
    r[0].hi[1] = inner.lo[1] - 1;                                               // rows below
    r[1].lo[1] = inner.hi[1] + 1;                                               // rows above
    r[2].lo[1] = inner.lo[1]; r[2].hi[1] = inner.hi[1]; r[2].hi[0] = inner.lo[0] - 1;   // columns left
    r[3].lo[1] = inner.lo[1]; r[3].hi[1] = inner.hi[1]; r[3].lo[0] = inner.hi[0] + 1;   // columns right
    for (rh_box &q : r) {
        q = bx_and(q, all);
        if (!bx_ok(q)) continue;
        k_zero_box<<<grid3(q), rh_block(), 0, c->stream>>>(mkB(q), mkA(a));
        RH_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::setup_step, REMORA_setup_step.cpp:7-369 (+ REMORA::prestep, REMORA_prestep.cpp:36-185)
extern "C" int rh_setup_step(rh_ctx *c, rh_state *s, int istep, double t_old)
{
    TScope tsc(c, "setup_step");
    const rh_params &P = c->prm;
    const int N = c->N, iic = istep, ntfirst = 0, nrhs = 0;
    const double dt = P.dt;
    const rh_box bx = c->bx;
    c->P_ready = false;
    if (c->poison) { RH_HIP(hipMemsetAsync(c->d_work, 0xFF, c->work_bytes, c->stream)); }
    {
        TScope t(c, "halo");
        BcScope bcs(c);
        RH_TRY(halo_fill_boundary(c, mfv(c, s->cons_old, 0, 0, 0), 0, RH_NCONS));                  // :25-28
        RH_TRY(halo_fill_boundary(c, mfv(c, s->xvel_old, 1, 0, 0), 0, 1));
        RH_TRY(halo_fill_boundary(c, mfv(c, s->yvel_old, 0, 1, 0), 0, 1));
        RH_TRY(halo_fillpatch(c, mfv(c, s->zvel_old, 0, 0, 1), s->mskr, RH_BCV_ZVEL, 0, 1));
        RH_TRY(halo_fill_boundary(c, mfv(c, s->cons_new, 0, 0, 0), 0, RH_NCONS));                  // :95-97
        RH_TRY(halo_fill_boundary(c, mfv(c, s->xvel_new, 1, 0, 0), 0, 1));
        RH_TRY(halo_fill_boundary(c, mfv(c, s->yvel_new, 0, 1, 0), 0, 1));
        RH_TRY(bcs.end());
    }
    RH_TRY(dev_setval(c, s->rufrc, 0, s->rufrc.ncomp, 0.0)); RH_TRY(dev_setval(c, s->rvfrc, 0, s->rvfrc.ncomp, 0.0));
    if (iic == ntfirst) {                                                                          // :105-110
        RH_TRY(dev_copy(c, s->cons_new, s->cons_old)); RH_TRY(dev_copy(c, s->xvel_new, s->xvel_old));
        RH_TRY(dev_copy(c, s->yvel_new, s->yvel_old)); RH_TRY(dev_copy(c, s->zvel_new, s->zvel_old));
    }
    RH_TRY(set_smflux(c, s, t_old));                                                               // :111
    {
        TScope t(c, "bstress");
        RH_TRY(launch_bstress(c, s));                                                              // :115-140
        BcScope bcs(c);
        RH_TRY(halo_fillpatch(c, mfv(c, s->bustr, 1, 0, 0), s->msku, RH_BCV_U2D_SIMPLE, 0, 1));
        RH_TRY(halo_fillpatch(c, mfv(c, s->bvstr, 0, 1, 0), s->mskv, RH_BCV_V2D_SIMPLE, 0, 1));
        RH_TRY(bcs.end());
    }
    // :187-197.  With the predictor on, the old mass fluxes are formed by the kernel that integrates omega from them (below): nothing
    // in between reads Huon / Hvom
    const bool defer_mf = P.use_prestep && !c->unfused;
    if (!defer_mf) { TScope t(c, "massflux_old"); RH_TRY(launch_massflux_old(c, s)); }
    // :200 rho_eos.  Its operands (cons_old, Hz, z_r, z_w, h, mskr) are not written before prsgrd (:262-266), and its results (bvf,
    // rhoA, rhoS) are first read after it (gls_prestep, the barotropic steps): the fused form runs both at prsgrd's place
    const bool rho_late = !c->unfused && !c->no_rhofuse;
    if (c->unfused) RH_TRY(rh_rho_eos(c, bx_grow(bx, 2, 2, 0), s->cons_old, c->rho, s->rhoA, s->rhoS, s->bvf, s->Hz, s->z_w, s->z_r, s->h, s->mskr, N));
    else if (!rho_late) RH_TRY(launch_rho_eos_P(c, s, bx_grow(bx, 2, 2, 0), bx_grow(bx, 1, 1, 0)));     // + the hydrostatic pressure of prsgrd (:229-260), same operands
    if (P.vmix_mode != RH_VMIX_GLS) RH_TRY(set_analytical_vmix(c, s));                             // :202-205
    RH_TRY(launch_set_zeta(c, s));                                                                 // :208
    // :210-211 mf_W.setVal(0): omega overwrites every level of the grow-1 box, so only the frame around it needs the zero
    if (P.use_prestep) RH_TRY(zero_outside(c, c->W, bx_grow(bx, 1, 1, 0))); else RH_TRY(dev_setval(c, c->W, 0, 1, 0.0));
    bool fused_t = false;
    if (P.use_prestep) {                                                                           // :216, REMORA_prestep.cpp
        int mo = 0;
        if (defer_mf) RH_TRY(launch_massflux_omega(c, s, bx_grow(bx, 1, 1, 0), N, &mo));
        if (!mo) {
            if (defer_mf) { TScope t(c, "massflux_old"); RH_TRY(launch_massflux_old(c, s)); }
            RH_TRY(rh_omega(c, bx_grow(bx, 1, 1, 0), c->W, s->Huon, s->Hvom, s->z_w, s->h, N));    // prestep_t_advection.cpp:97-125 (identical for all tracers)
        }
        fused_t = !c->unfused;
        if (fused_t) RH_TRY(launch_prestep_t3(c, s, iic, ntfirst));             // 3 tracers x (prestep_t + t3dmix of :283-286) in one sweep
        else
            for (int n = 0; n < RH_NCONS; n++)
                RH_TRY(rh_prestep_t(c, bx, a4_comp(s->cons_old, n), a4_comp(s->cons_new, n), a4_comp(s->sstore, n), s->Hz, s->Huon, s->Hvom, c->W,
                                    s->pm, s->pn, s->msku, s->mskv, iic, ntfirst, N, dt));
        const rh_box tbxp1 = bx_grow(bx, 1, 1, 0);
        // ru / rv(nrhs) on the valid faces are assigned by prsgrd below before anything reads them: prestep_uv skips that store there
        const rh_box none = bx_make(0, 0, 0, -1, -1, -1);
        if (c->unfused) {
            RH_TRY(launch_prestep_uv(c, tbxp1, none, 1, 0, s->xvel_new, s->xvel_old, s->ru, s->Hz, s->sustr, s->bustr, s->pm, s->pn, iic, ntfirst, nrhs, N, dt));
            RH_TRY(launch_prestep_uv(c, tbxp1, none, 0, 1, s->yvel_new, s->yvel_old, s->rv, s->Hz, s->svstr, s->bvstr, s->pm, s->pn, iic, ntfirst, nrhs, N, dt));
        } else RH_TRY(launch_prestep_uv2(c, s, tbxp1, bx_nodal(bx, 0), bx_nodal(bx, 1), iic, ntfirst, nrhs, N, dt));
    }
    {
        TScope t(c, "halo");
        BcScope bcs(c);
        const MFV mW = mfv(c, c->W, 0, 0, 1);
        RH_TRY(halo_fill_boundary(c, mW, 0, 1));                                                   // :226-227
        RH_TRY(halo_physbc(c, mW, s->mskr, 0, 1, RH_BCV_ZVEL, 0));
        RH_TRY(bcs.end());
    }
    const rh_box xbx = bx_nodal(bx, 0), ybx = bx_nodal(bx, 1);
    int rp = 0;
    if (rho_late) RH_TRY(launch_rho_prsgrd(c, s, bx_grow(bx, 2, 2, 0), xbx, ybx, nrhs, N, &rp));
    if (!rp) {
        if (rho_late) RH_TRY(launch_rho_eos_P(c, s, bx_grow(bx, 2, 2, 0), bx_grow(bx, 1, 1, 0)));
        RH_TRY(rh_prsgrd(c, bx_grow(bx, 1, 1, 0), xbx, ybx, s->ru, s->rv, s->pn, s->pm, c->rho, c->P, s->Hz, s->z_r, s->z_w, s->msku, s->mskv, nrhs, N));
    }
    if (!fused_t) {
        RH_TRY(rh_t3dmix(c, bx, s->cons_new, s->cons_old, s->diff2, s->Hz, s->pm, s->pn, s->msku, s->mskv, dt, P.use_salt ? 2 : 1));   // :283-286
        RH_TRY(rh_t3dmix(c, bx, a4_comp(s->cons_new, 2), a4_comp(s->cons_old, 2), a4_comp(s->diff2, 2), s->Hz, s->pm, s->pn, s->msku, s->mskv, dt, 1));
    }
    RH_TRY(rh_rhs_uv_3d(c, xbx, ybx, s->xvel_old, s->yvel_old, s->ru, s->rv, s->rufrc, s->rvfrc, s->sustr, s->svstr, s->bustr, s->bvstr,
                        s->Huon, s->Hvom, s->pm, s->pn, c->W, s->Hz, s->fcor, P.use_coriolis, nrhs, N));
    if (P.use_uv3dmix)
        RH_TRY(rh_uv3dmix(c, xbx, ybx, s->xvel_new, s->yvel_new, s->xvel_old, s->yvel_old, s->rufrc, s->rvfrc, s->visc2_p, s->visc2_r, s->Hz,
                          s->pm, s->pn, s->mskp, nrhs, 0, dt));
    if (P.vmix_mode == RH_VMIX_GLS) RH_TRY(rh_gls_prestep(c, s, c->W, iic % 2, (iic + 1) % 2, iic, ntfirst));   // :352-357
    // The ghost cells filled here are first read in advance_3d, a whole barotropic loop later: inside rh_advance the fills go to
    // the side stream (pack, send/recv, unpack and BC passes overlap the 2-D kernels) and rh_advance_3d_ml joins them.
    const bool side = c->in_advance && !c->no_async;
    AsyncScope as(side ? c : nullptr);                      // swaps back (and records the join event) on every exit path
    if (side) RH_TRY(as.begin());
    {
        TScope t(c, "halo");
        BcScope bcs(c);
        RH_TRY(halo_fillpatch(c, mfv(c, s->sstore, 0, 0, 0), s->mskr, RH_BCV_CONS, 0, RH_NCONS, &s->cons_old, &s->msku, &s->mskv));  // :363
        RH_TRY(halo_fill_boundary(c, mfv(c, s->Huon, 1, 0, 0), 0, 1));                             // :367-368
        RH_TRY(halo_fill_boundary(c, mfv(c, s->Hvom, 0, 1, 0), 0, 1));
        RH_TRY(bcs.end());
    }
    if (side) RH_TRY(as.end());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// REMORA::advance_3d, REMORA_advance_3d.cpp:10-387
static int advance_3d(rh_ctx *c, rh_state *s, int istep)
{
    const int N = c->N; const double dt = c->prm.dt;
    const rh_box bx = c->bx, xbx = bx_nodal(bx, 0), ybx = bx_nodal(bx, 1);
    const rh_box gbx2 = bx_grow(bx, 2, 2, 0), gbx1 = bx_grow(bx, 1, 1, 0);
    RH_TRY(rh_stretch_transform(c, s));                                                            // :40
    int fused_uv = 0;
    RH_TRY(launch_uv_corrector(c, s, istep, &fused_uv));                                           // :104-150 in one column kernel per velocity
    if (!fused_uv) {
        RH_TRY(launch_uv_ab3(c, s, istep));                                                        // :104-123
        RH_TRY(vert_visc_3d_view(c, xbx, 1, 0, s->xvel_new, s->Hz, s->Akv, mkA_vmix(c, s, 0), N, dt));   // :132
        RH_TRY(vert_visc_3d_view(c, ybx, 0, 1, s->yvel_new, s->Hz, s->Akv, mkA_vmix(c, s, 0), N, dt));   // :138
        RH_TRY(rh_vert_mean_3d(c, xbx, 1, 0, s->xvel_new, s->Hz, s->DU_avg1, s->pn, s->msku, N));  // :144
        RH_TRY(rh_vert_mean_3d(c, ybx, 0, 1, s->yvel_new, s->Hz, s->DV_avg1, s->pm, s->mskv, N));  // :150
    }
    {
        TScope t(c, "halo");
        BcScope bcs(c);
        RH_TRY(halo_physbc(c, mfv(c, s->xvel_new, 1, 0, 0), s->msku, 0, 1, RH_BCV_XVEL, 0, &s->xvel_old));       // :154-155
        RH_TRY(halo_physbc(c, mfv(c, s->yvel_new, 0, 1, 0), s->mskv, 0, 1, RH_BCV_YVEL, 0, &s->yvel_old));
        RH_TRY(bcs.end());
    }
    if (fused_uv == 2) {                                   // the corrector's sweep did the valid faces; ghost and wall faces remain
        RH_TRY(launch_massflux_rest(c, gbx2, xbx, 1, 0, s->xvel_new, s->ubar, s->Huon, s->Hz, s->pn, s->DU_avg1, s->DU_avg2, s->msku, N));   // :213
        RH_TRY(launch_massflux_rest(c, gbx2, ybx, 0, 1, s->yvel_new, s->vbar, s->Hvom, s->Hz, s->pm, s->DV_avg1, s->DV_avg2, s->mskv, N));   // :218
    } else {
        RH_TRY(rh_update_massflux_3d(c, gbx2, 1, 0, s->xvel_new, s->ubar, s->Huon, s->Hz, s->pn, s->DU_avg1, s->DU_avg2, s->msku, N));   // :213
        RH_TRY(rh_update_massflux_3d(c, gbx2, 0, 1, s->yvel_new, s->vbar, s->Hvom, s->Hz, s->pm, s->DV_avg1, s->DV_avg2, s->mskv, N));   // :218
    }
    RH_TRY(zero_outside(c, c->W, gbx1));                                                           // :230-231 (omega overwrites gbx1)
    RH_TRY(rh_omega(c, gbx1, c->W, s->Huon, s->Hvom, s->z_w, s->h, N));                            // :272-293
    if (c->prm.vmix_mode == RH_VMIX_GLS) RH_TRY(rh_gls_corrector(c, s, c->W, istep % 2, 1 - istep % 2));   // :296-303
    if (!c->unfused) RH_TRY(launch_rhs_t3(c, s));                                                  // :341-346, 3 tracers in one sweep
    else for (int n = 0; n < RH_NCONS; n++)
        RH_TRY(rh_rhs_t_3d(c, bx, a4_comp(s->cons_new, n), a4_comp(s->sstore, n), s->Huon, s->Hvom, s->Hz, s->pn, s->pm, c->W, s->msku, s->mskv, N, dt));
    { TScope t(c, "halo"); RH_TRY(halo_fillpatch(c, mfv(c, s->cons_new, 0, 0, 0), s->mskr, RH_BCV_CONS, 0, RH_NCONS, &s->cons_old, &s->msku, &s->mskv)); }   // :350
    for (int n = 0; n < RH_NCONS; n++)                                                             // :382-385
        RH_TRY(vert_visc_3d_view(c, bx, 0, 0, a4_comp(s->cons_new, n), s->Hz, a4_comp(s->Akt, n), mkA_vmix(c, s, 1 + n), N, dt));
    return 0;
}

// REMORA::advance_3d_ml, REMORA_advance_3d_ml.cpp:6-71
extern "C" int rh_advance_3d_ml(rh_ctx *c, rh_state *s, int istep)
{
    RH_TRY(halo_async_join(c));                            // the fills setup_step left on the side stream
    TScope tsc(c, "advance_3d_ml");
    {
        TScope t(c, "halo");
        BcScope bcs(c);
        RH_TRY(halo_fill_boundary(c, mfv(c, s->xvel_new, 1, 0, 0), 0, 1));                         // :11-12
        RH_TRY(halo_fill_boundary(c, mfv(c, s->yvel_new, 0, 1, 0), 0, 1));
        RH_TRY(bcs.end());
    }
    RH_TRY(advance_3d(c, s, istep));
    {
        TScope t(c, "halo");
        BcScope bcs(c);
        RH_TRY(halo_fill_boundary(c, mfv(c, s->ubar, 1, 0, 0), 0, 3));                             // :33-38
        RH_TRY(halo_fill_boundary(c, mfv(c, s->vbar, 0, 1, 0), 0, 3));
        RH_TRY(halo_fillpatch(c, mfv(c, s->cons_new, 0, 0, 0), s->mskr, RH_BCV_CONS, 0, RH_NCONS, &s->cons_old, &s->msku, &s->mskv)); // :45
        RH_TRY(halo_fill_boundary(c, mfv(c, s->xvel_new, 1, 0, 0), 0, 1));                         // :46-47
        RH_TRY(halo_fill_boundary(c, mfv(c, s->yvel_new, 0, 1, 0), 0, 1));
        RH_TRY(halo_fillpatch(c, mfv(c, s->zvel_new, 0, 0, 1), s->mskr, RH_BCV_ZVEL, 0, 1));       // :48
        RH_TRY(bcs.end());
    }
    if (!c->prm.masks_all_one) RH_TRY(launch_mask_cons(c, s));                                     // :51-62 (x *= 1.0 is the identity)
    return 0;
}

// REMORA::Advance, REMORA_Advance.cpp:7-35
extern "C" int rh_advance(rh_ctx *c, rh_state *s, int istep, double t_old)
{
    c->in_advance = true;
    const int rc0 = rh_setup_step(c, s, istep, t_old);
    c->in_advance = false;
    RH_TRY(rc0);
    if (c->prm.use_barotropic) {
        const int nfast_counter = c->nfast + 1;
        // inside this loop a decomposed run may leave the new level's ghost fill owed from one call to the next (rh_step2d.hip);
        // rh_advance_3d_ml's join settles the last one
        c->in_loop2d = true;
        int rc = 0;
        for (int my_iif = 0; my_iif < nfast_counter && !rc; my_iif++) rc = rh_advance_2d_onestep(c, s, istep, my_iif);
        c->in_loop2d = false;
        if (rc) { c->pend2d.on = false; return rc; }
    }
    return rh_advance_3d_ml(c, s, istep);
}

// ------------------------------------------------------------------------------------------------------------------
// Synthetic initial state: REMORA::init_only (Source/REMORA.cpp:608-702), set_2darrays (Initialization/REMORA_init.cpp:97-154),
// set_pm_pn (REMORA_make_new_level.cpp:548-553), init_custom_{bathymetry,prob,hmix,smflux} of Exec/<case>/prob.cpp
struct InitPar { int prob, nx, ny, perx, pery, flat, use_salt; double plo[3], phi[3], dx0, dx1, T0, S0, rho0, g, u0, v0, z0, zRef, uRef; };

__global__ void k_init_h(B3 b, A4 h, InitPar q)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    if (q.prob == 0) {                                                         // Exec/Upwelling/prob.cpp:21-80
        const int Lm = q.nx, Mm = q.ny;
        if (q.pery) {
            const int iFort = i + 1; double val1 = (iFort <= Lm / 2.0) ? iFort : Lm + 1 - iFort; val1 -= 0.5;
            const double adj = q.dx1 / 1000.0;
            h(i, j, 0, 0) = dmin_(-q.plo[2], (84.5 + 66.526 * tanh((val1 * adj - 10.0) / 7.0)));
        } else if (q.perx) {
            const int jFort = j + 1; double val1 = (jFort <= Mm / 2.0) ? jFort : Mm + 1 - jFort; val1 -= 0.5;
            const double adj = q.dx0 / 1000.0;
            h(i, j, 0, 0) = dmin_(-q.plo[2], (84.5 + 66.526 * tanh((val1 * adj - 10.0) / 7.0)));
        }
    } else if (q.prob == 1) {                                                  // Exec/Seamount/prob.cpp:25-72
        const double Xsize = 320000.0, Esize = 320000.0, depth = 5000.0;
        if (!q.flat) {
            const double x = q.plo[0] + (i + 0.5) * q.dx0, y = q.plo[1] + (j + 0.5) * q.dx1;
            const double val1 = (x - 0.5 * Xsize) / 40000.0, val2 = (y - 0.5 * Esize) / 40000.0;
            h(i, j, 0, 0) = depth - 4500.0 * exp(-(val1 * val1 + val2 * val2));
        } else { h(i, j, 0, 0) = -q.plo[2]; h(i, j, 0, 1) = -q.plo[2]; }
    } else if (q.prob == 3) { h(i, j, 0, 0) = -q.plo[2]; h(i, j, 0, 1) = -q.plo[2]; }   // Exec/Advection/prob.cpp:29-52
    else if (q.prob == 4) {                                                    // Exec/DoublyPeriodic/prob.cpp:32-105
        const int Lm = q.nx, Mm = q.ny;
        if (q.flat) { h(i, j, 0, 0) = -q.plo[2]; h(i, j, 0, 1) = -q.plo[2]; }
        else if (q.pery) {
            const int iFort = i + 1; const double val1 = (iFort <= Lm / 2.0) ? iFort : Lm + 1 - iFort;
            h(i, j, 0, 0) = dmin_(-q.plo[2], (84.5 + 66.526 * tanh((val1 - 10.0) / 7.0)));
        } else if (q.perx) {
            const int jFort = j + 1; const double val1 = (jFort <= Mm / 2.0) ? jFort : Mm + 1 - jFort;
            h(i, j, 0, 0) = dmin_(-q.plo[2], (84.5 + 66.526 * tanh((val1 - 10.0) / 7.0)));
        }
    }
}
__global__ void k_init_cons(B3 b, A4 st, A4 z_r, InitPar q)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const int k = b.lo2 + RH_KZ;
    const double z = z_r(i, j, k);
    if (q.prob == 0) { st(i, j, k, 0) = q.T0 + 8.0 * exp(z / 50.0); if (q.use_salt) st(i, j, k, 1) = q.S0; st(i, j, k, 2) = 0.0; }          // Upwelling/prob.cpp:101-161
    else if (q.prob == 1) { st(i, j, k, 0) = q.T0 + 7.5 * exp(z / 1000.0); if (q.use_salt) st(i, j, k, 1) = q.S0; st(i, j, k, 2) = 0.0; } // Seamount/prob.cpp:93-165
    else if (q.prob == 2) {                                                    // DoubleGyre/prob.cpp:49-114
        const double val1 = (44.69 / 39.382) * (44.69 / 39.382);
        const double val2 = val1 * (q.rho0 * 100.0 / q.g) * (5.0e-5 / ((42.689 / 44.69) * (42.689 / 44.69)));
        const double y = q.plo[1] + (j + 0.5) * q.dx1, yextent = q.phi[1] - q.plo[1];
        const double val3 = q.T0 + val2 * exp(z / 100.0) * (10.0 - 0.4 * tanh(z / 100.0));
        const double val4 = y / yextent;
        st(i, j, k, 0) = val3 - 3.0 * val4;
        if (q.use_salt) st(i, j, k, 1) = 34.5 - 0.001 * z - val4;
        st(i, j, k, 2) = 0.0;
    } else if (q.prob == 5) {                                                  // Channel_Test/prob.cpp:73-110
        const double y = q.plo[1] + (j + 0.5) * q.dx1;
        st(i, j, k, 0) = q.T0 + z / (9.8 * 1.7);
        if (q.use_salt) st(i, j, k, 1) = q.S0 + y / (9.8 * 760.0);
        st(i, j, k, 2) = 0.0;
    } else if (q.prob == 4) {                                                  // DoublyPeriodic/prob.cpp:126-180
        st(i, j, k, 0) = q.T0 + 8.0 * exp(z / 50.0);
        if (q.use_salt) st(i, j, k, 1) = q.S0;
        const double xcent = 0.5 * (q.plo[0] + q.phi[0]), ycent = 0.5 * (q.plo[1] + q.phi[1]);
        const double x = q.plo[0] + (i + 0.5) * q.dx0 - xcent, y = q.plo[1] + (j + 0.5) * q.dx1 - ycent;
        const double r2 = x * x + y * y, rad = 0.1 * (q.phi[0] - q.plo[0]), radsq = rad * rad;
        st(i, j, k, 2) = (r2 < radsq) ? 1.0 : 0.0;
    } else {                                                                   // Advection/prob.cpp:72-157
        st(i, j, k, 0) = q.T0;
        const double xcent = 0.5 * (q.plo[0] + q.phi[0]), ycent = 0.5 * (q.plo[1] + q.phi[1]);
        const double x = q.plo[0] + (i + 0.5) * q.dx0 - xcent, y = q.plo[1] + (j + 0.5) * q.dx1 - ycent;
        const double r2 = x * x + y * y, rad = 0.1 * (q.phi[0] - q.plo[0]), radsq = rad * rad;
        if (q.use_salt) st(i, j, k, 1) = q.S0;
        st(i, j, k, 2) = exp(-r2 / (2. * radsq));
    }
}
// log-layer x-velocity of Exec/DoublyPeriodic/prob.cpp:182-198 (z_r of the cell to the right of the face, as the reference indexes it)
__global__ void k_init_u_log(B3 b, A4 u, A4 z_r, InitPar q)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const int k = b.lo2 + RH_KZ;
    const double z = -z_r(i, j, k);
    u(i, j, k) = q.u0 + q.uRef * log((z + q.z0) / q.z0) / log((q.zRef + q.z0) / q.z0);
}
// sheared, wavy along-channel flow of Exec/Channel_Test/prob.cpp:126-141 (x, y at the cell centre of index (i, j), as the reference has it)
__global__ void k_init_u_channel(B3 b, A4 u, A4 z_r, InitPar q)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const int k = b.lo2 + RH_KZ;
    const double z = z_r(i, j, k), depth = 50.0;
    const double x = q.plo[0] + (i + 0.5) * q.dx0, y = q.plo[1] + (j + 0.5) * q.dx1;
    const double val1 = x * 6.28318530718 * 1.0e-5, val2 = y * 6.28318530718 * 1.0e-5;
    u(i, j, k) = (depth + z) * 0.01 + 0.05 * sin(val1) * sin(val2);
}
__global__ void k_setbox(B3 b, A4 a, double v)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const int k = b.lo2 + RH_KZ;
    a(i, j, k) = v;
}
// depth-mean velocity, Initialization/REMORA_init.cpp:97-154
__global__ void k_set2d(B3 ub, B3 vb, B3 un, A4 ubar, A4 vbar, A4 Hz, A4 u, A4 v, int N)
{
    RH_IJ(un);
    if (inb2(ub, i, j)) {
        double CF = 0., sum = 0.;
        for (int k = 0; k <= N; k++) { const double a = 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k)); sum += a; CF += a * u(i, j, k); }
        ubar(i, j, 0, 0) = CF / sum;
    }
    if (inb2(vb, i, j)) {
        double CF = 0., sum = 0.;
        for (int k = 0; k <= N; k++) { const double a = 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k)); sum += a; CF += a * v(i, j, k); }
        vbar(i, j, 0, 0) = CF / sum;
    }
}
__global__ void k_fcor(B3 b, A4 fcor, double f0, double beta, double plo1, double dx1, double Esize)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const double y = plo1 + (j + 0.5) * dx1;
    fcor(i, j, 0) = f0 + beta * (y - 0.5 * Esize);
}
__global__ void k_wind_dg(B3 b, A4 sustr, double windamp, double val1, double plo1, double dx1)   // DoubleGyre/prob.cpp:167-206
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const double y = plo1 + (j + 0.5) * dx1;
    sustr(i, j, 0) = windamp * cos(val1 * y);
}
__global__ void k_copy_comp2d(B3 b, A4 dst, A4 src)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    dst(i, j, 0) = src(i, j, 0);
}

int dev_setbox(rh_ctx *c, const rh_array4 &a, int comp, rh_box b, double v)
{
    b = bx_and(b, bx_of(a)); if (!bx_ok(b)) return 0;
    k_setbox<<<grid3(b), rh_block(), 0, c->stream>>>(mkB(b), mkA(a, comp), v);
    RH_LAUNCH_CHECK(); return 0;
}

extern "C" int rh_init_state(rh_ctx *c, rh_state *s, const rh_init_params *ip)
{
    const rh_params &P = c->prm;
    c->vmix_akv_done = c->vmix_akt_done = nullptr;          // Akv / Akt are rewritten below: the next step forms the constants again
    const int N = c->N;
    const rh_box bx = c->bx, b2 = bx_slab(bx, 2, 0);
    if (ip->prob < 0 || ip->prob > 5) return rh_fail("rh_init_state: unknown problem %d", ip->prob);
    InitPar q; memset(&q, 0, sizeof q);
    q.prob = ip->prob; q.nx = c->dom.hi[0] - c->dom.lo[0] + 1; q.ny = c->dom.hi[1] - c->dom.lo[1] + 1;
    q.perx = P.periodic[0]; q.pery = P.periodic[1]; q.flat = P.flat_bathymetry; q.use_salt = P.use_salt;
    for (int d = 0; d < 3; d++) { q.plo[d] = P.prob_lo[d]; q.phi[d] = P.prob_hi[d]; }
    q.dx0 = (P.prob_hi[0] - P.prob_lo[0]) / q.nx; q.dx1 = (P.prob_hi[1] - P.prob_lo[1]) / q.ny;
    q.T0 = P.T0; q.S0 = P.S0; q.rho0 = P.rho0; q.g = P.g; q.u0 = ip->u_0; q.v0 = ip->v_0; q.z0 = ip->z0; q.zRef = ip->zRef; q.uRef = ip->uRef;
    const double dxi0 = q.nx / (P.prob_hi[0] - P.prob_lo[0]), dxi1 = q.ny / (P.prob_hi[1] - P.prob_lo[1]);

    // everything starts at zero (the reference poisons fresh FABs with signalling NaNs and then sets what it reads)
    {
        rh_array4 *all[] = {&s->cons_old, &s->cons_new, &s->xvel_old, &s->xvel_new, &s->yvel_old, &s->yvel_new, &s->zvel_old, &s->zvel_new,
            &s->Zt_avg1, &s->h, &s->ubar, &s->vbar, &s->ru, &s->rv, &s->ru2d, &s->rv2d, &s->mskr, &s->msku, &s->mskv, &s->mskp,
            &s->z_w, &s->z_r, &s->Hz, &s->Huon, &s->Hvom, &s->Akv, &s->Akt, &s->visc2_p, &s->visc2_r, &s->diff2,
            &s->rufrc, &s->rvfrc, &s->sustr, &s->svstr, &s->rdrag, &s->bustr, &s->bvstr, &s->DU_avg1, &s->DU_avg2, &s->DV_avg1, &s->DV_avg2,
            &s->rubar, &s->rvbar, &s->rzeta, &s->zeta, &s->pm, &s->pn, &s->fcor, &s->sstore, &s->rhoS, &s->rhoA, &s->bvf};
        for (rh_array4 *a : all) RH_TRY(dev_setval(c, *a, 0, a->ncomp, 0.0));
        if (P.vmix_mode == RH_VMIX_GLS)
            for (rh_array4 *a : {&s->tke, &s->gls, &s->Lscale, &s->Akk, &s->Akp}) {
                if (!a->p) return rh_fail("rh_init_state: vmix_mode = RH_VMIX_GLS needs the state's tke / gls / Lscale / Akk / Akp arrays");
                RH_TRY(dev_setval(c, *a, 0, a->ncomp, 0.0));
            }
    }
    for (rh_array4 *m : {&s->mskr, &s->msku, &s->mskv, &s->mskp}) RH_TRY(dev_setval(c, *m, 0, 1, 1.0));   // init_masks, make_new_level.cpp:381-389
    RH_TRY(dev_setval(c, s->rdrag, 0, 1, ip->rdrag));
    // bathymetry (set_bathymetry, REMORA.cpp:464-526)
    if (ip->prob == 2) RH_TRY(dev_setval(c, s->h, 0, 2, 500.0));                                   // DoubleGyre/prob.cpp:21-28
    else if (ip->prob == 5) RH_TRY(dev_setval(c, s->h, 0, 2, 50.0));                               // Channel_Test/prob.cpp:20-52
    else {
        RH_TRY(dev_setval(c, s->h, 0, 2, P.prob_hi[2]));
        rh_box hb = (ip->prob == 0 || ip->prob == 4) ? bx_grow(b2, 2, 2, 0) : (ip->prob == 1 ? bx_grow(b2, 3, 3, 0) : bx_of(s->h));
        hb = bx_and(hb, bx_of(s->h));
        k_init_h<<<grid2(hb), rh_block(), 0, c->stream>>>(mkB(hb), mkA(s->h), q);
        RH_LAUNCH_CHECK();
    }
    RH_TRY(halo_fill_boundary(c, mfv(c, s->h, 0, 0, 0), 0, 2));
    RH_TRY(halo_fill_boundary(c, mfv(c, s->zeta, 0, 0, 0), 0, 3));                                 // set_zeta: zeta = 0, Zt_avg1 = zeta
    RH_TRY(rh_stretch_transform(c, s));
    // init_custom_prob
    k_init_cons<<<grid3(bx), rh_block(), 0, c->stream>>>(mkB(bx), mkA(s->cons_new), mkA(s->z_r), q);
    RH_LAUNCH_CHECK();
    if (ip->prob == 4 || ip->prob == 5) {
        const rh_box xb = bx_nodal(bx, 0);
        RH_NEED(s->xvel_new, xb); RH_NEED(s->z_r, xb);
        if (ip->prob == 4) k_init_u_log<<<grid3(xb), rh_block(), 0, c->stream>>>(mkB(xb), mkA(s->xvel_new), mkA(s->z_r), q);
        else k_init_u_channel<<<grid3(xb), rh_block(), 0, c->stream>>>(mkB(xb), mkA(s->xvel_new), mkA(s->z_r), q);
        RH_LAUNCH_CHECK();
    } else RH_TRY(dev_setbox(c, s->xvel_new, 0, bx_nodal(bx, 0), ip->prob == 3 ? ip->u_0 : 0.0));
    RH_TRY(dev_setbox(c, s->yvel_new, 0, bx_nodal(bx, 1), (ip->prob == 3 || ip->prob == 4) ? ip->v_0 : 0.0));
    RH_TRY(dev_setval(c, s->pm, 0, 1, dxi0)); RH_TRY(dev_setval(c, s->pn, 0, 1, dxi1));            // set_pm_pn
    {                                                                                              // set_2darrays
        const rh_box ub = bx_grow(bx_nodal(b2, 0), 2, 2, 0), vb = bx_grow(bx_nodal(b2, 1), 2, 2, 0), un = bx_union(ub, vb);
        RH_NEED(s->ubar, ub); RH_NEED(s->vbar, vb); RH_NEED(s->Hz, bx_growlo_i(bx_grow(bx_nodal(bx, 0), 2, 2, 0))); RH_NEED(s->Hz, bx_growlo_j(bx_grow(bx_nodal(bx, 1), 2, 2, 0)));
        RH_NEED(s->xvel_new, bx_grow(bx_nodal(bx, 0), 2, 2, 0)); RH_NEED(s->yvel_new, bx_grow(bx_nodal(bx, 1), 2, 2, 0));
        k_set2d<<<grid2(un), rh_block(), 0, c->stream>>>(mkB(ub), mkB(vb), mkB(un), mkA(s->ubar), mkA(s->vbar), mkA(s->Hz), mkA(s->xvel_new), mkA(s->yvel_new), N);
        RH_LAUNCH_CHECK();
        RH_TRY(halo_fillpatch(c, mfv(c, s->ubar, 1, 0, 0), s->msku, RH_BCV_UBAR, 0, 1));
        RH_TRY(halo_fillpatch(c, mfv(c, s->vbar, 0, 1, 0), s->mskv, RH_BCV_VBAR, 0, 1));
    }
    if (P.vmix_mode == RH_VMIX_GLS) RH_TRY(launch_gls_init(c, s)); else RH_TRY(set_analytical_vmix(c, s));   // init_set_vmix, REMORA.cpp:550-562
    {                                                                                              // set_hmixcoef, REMORA.cpp:575-596
        double v = 0.0, d[RH_NCONS] = {0.0, 0.0, 0.0};
        if (ip->hmix_constant) { v = ip->visc2; for (int n = 0; n < RH_NCONS; n++) d[n] = ip->tnu2[n]; }
        else if (ip->prob == 0 || ip->prob == 3 || ip->prob == 4) v = 5.0;
        else if (ip->prob == 5) v = 2.0;                                                           // Channel_Test/prob.cpp:178-202
        else if (ip->prob == 2) { v = 1280.0; for (int n = 0; n < RH_NCONS; n++) d[n] = 1280.0; }
        RH_TRY(dev_setval(c, s->visc2_p, 0, 1, v)); RH_TRY(dev_setval(c, s->visc2_r, 0, 1, v));
        for (int n = 0; n < RH_NCONS; n++) RH_TRY(dev_setval(c, s->diff2, n, 1, d[n]));
        RH_TRY(halo_fillpatch(c, mfv(c, s->visc2_p, 0, 0, 0), s->mskr, RH_BCV_FOEXTRAP_PERIODIC, 0, 1));
        RH_TRY(halo_fillpatch(c, mfv(c, s->visc2_r, 0, 0, 0), s->mskr, RH_BCV_FOEXTRAP_PERIODIC, 0, 1));
        for (int n = 0; n < RH_NCONS; n++) RH_TRY(halo_fillpatch(c, mfv(c, s->diff2, 0, 0, 0), s->mskr, RH_BCV_FOEXTRAP_PERIODIC, n, 1));
    }
    if (P.use_coriolis) {                                                                          // set_coriolis, REMORA_init.cpp:48-74
        const rh_box fb = bx_of(s->fcor);
        k_fcor<<<grid2(fb), rh_block(), 0, c->stream>>>(mkB(fb), mkA(s->fcor), ip->coriolis_f0, ip->coriolis_beta, P.prob_lo[1], q.dx1, P.prob_hi[1] - P.prob_lo[1]);
        RH_LAUNCH_CHECK();
        const MFV mf = mfv(c, s->fcor, 0, 0, 0);
        RH_TRY(halo_fill_boundary(c, mf, 0, 1));
        RH_TRY(halo_fillpatch(c, mf, s->mskr, RH_BCV_FOEXTRAP, 0, 1));
        RH_TRY(halo_fill_boundary(c, mf, 0, 1));
    }
    if (ip->prob == 2) {                                                                           // init_custom_smflux
        const double pi = 3.14159265359, yextent = P.prob_hi[1] - P.prob_lo[1];
        const rh_box sb = bx_of(s->sustr);
        k_wind_dg<<<grid2(sb), rh_block(), 0, c->stream>>>(mkB(sb), mkA(s->sustr), -0.05 / P.rho0, 2.0 * pi / yextent, P.prob_lo[1], q.dx1);
        RH_LAUNCH_CHECK();
    } else RH_TRY(set_smflux(c, s, 0.0));
    // REMORA::InitData, Source/REMORA.cpp:293-304
    // REMORA.cpp:293-295: mf_calc is the array being filled
    RH_TRY(halo_fillpatch(c, mfv(c, s->cons_new, 0, 0, 0), s->mskr, RH_BCV_CONS, 0, RH_NCONS, &s->cons_new, &s->msku, &s->mskv));
    RH_TRY(halo_fillpatch(c, mfv(c, s->xvel_new, 1, 0, 0), s->msku, RH_BCV_XVEL, 0, 1, &s->xvel_new));
    RH_TRY(halo_fillpatch(c, mfv(c, s->yvel_new, 0, 1, 0), s->mskv, RH_BCV_YVEL, 0, 1, &s->yvel_new));
    RH_TRY(halo_fillpatch(c, mfv(c, s->zvel_new, 0, 0, 1), s->mskr, RH_BCV_ZVEL, 0, 1));
    RH_TRY(dev_copy(c, s->cons_old, s->cons_new)); RH_TRY(dev_copy(c, s->xvel_old, s->xvel_new));
    RH_TRY(dev_copy(c, s->yvel_old, s->yvel_new)); RH_TRY(dev_copy(c, s->zvel_old, s->zvel_new));
    RH_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// REMORA::update_mskp, Initialization/REMORA_make_new_level.cpp:617-655: the vorticity-point mask of a node from the four rho-point
// masks around it (1: all wet or one dry; 2: two dry cells side by side, a straight coast; 0: anything else), on the cell box grown
// by one; then the ghost cells of the four masks as REMORA::init_masks_from_netcdf leaves them (init_from_netcdf.cpp:408-412).
__global__ void k_update_mskp(B3 b, A4 mskr, A4 mskp)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const bool a = mskr(i - 1, j, 0) > 0.5, bb = mskr(i, j, 0) > 0.5, cc = mskr(i - 1, j - 1, 0) > 0.5, d = mskr(i, j - 1, 0) > 0.5;
    const bool a0 = mskr(i - 1, j, 0) < 0.5, b0 = mskr(i, j, 0) < 0.5, c0 = mskr(i - 1, j - 1, 0) < 0.5, d0 = mskr(i, j - 1, 0) < 0.5;
    double v = 0.0;
    if (a && bb && cc && d) v = 1.0;
    else if ((a0 && bb && cc && d) || (a && b0 && cc && d) || (a && bb && c0 && d) || (a && bb && cc && d0)) v = 1.0;
    else if ((a && b0 && cc && d0) || (a0 && bb && c0 && d) || (a && bb && c0 && d0) || (a0 && b0 && cc && d)) v = 2.0;
    mskp(i, j, 0) = v;
}
extern "C" int rh_update_mskp(rh_ctx *c, rh_state *s)
{
    const rh_box b = bx_grow(bx_slab(c->bx, 2, 0), 1, 1, 0);
    RH_NEED(s->mskp, b); RH_NEED(s->mskr, bx_grow(b, 1, 1, 0));
    // mskr's ghost cells at rank seams (and across periodic sides) first: the kernel reads mskr(i-1, j), (i, j-1), (i-1, j-1) on the
    // box grown by one, and its first valid node line at a seam must see the neighbour's cells -- in the reference the FAB
    // intersection copy of init_from_netcdf.cpp:400 has put them there before update_mskp runs.  (Across a PERIODIC side the
    // reference still has init_masks' 1.0 in those ghost cells at that point; here they hold the wrapped mask.  No reference case
    // has land: parity unpinned either way.)
    RH_TRY(halo_fill_boundary(c, mfv(c, s->mskr, 0, 0, 0), 0, 1));
    k_update_mskp<<<grid2(b), rh_block(), 0, c->stream>>>(mkB(b), mkA(s->mskr), mkA(s->mskp));
    RH_LAUNCH_CHECK();
    RH_TRY(halo_fill_boundary(c, mfv(c, s->msku, 1, 0, 0), 0, 1));
    RH_TRY(halo_fill_boundary(c, mfv(c, s->mskv, 0, 1, 0), 0, 1));
    RH_TRY(halo_fill_boundary(c, mfv(c, s->mskp, 1, 1, 0), 0, 1));
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// plotfile variables, IO/REMORA_Plotfile.cpp:185-235: temp, salt, [scalar], cell-centred averages of the face velocities
__global__ void k_plot(B3 b, A4 cons, A4 u, A4 v, A4 w, double *__restrict__ out, int with_scalar)
{
    RH_IJ(b); if (!inb2(b, i, j)) return;
    const int k = b.lo2 + RH_KZ;
    const long nx = b.hi0 - b.lo0 + 1, ny = b.hi1 - b.lo1 + 1, nz = b.hi2 - b.lo2 + 1, n3 = nx * ny * nz;
    const long q = (long)(i - b.lo0) + (long)(j - b.lo1) * nx + (long)(k - b.lo2) * nx * ny;
    int nc = 0;
    out[nc++ * n3 + q] = cons(i, j, k, 0);
    out[nc++ * n3 + q] = cons(i, j, k, 1);
    if (with_scalar) out[nc++ * n3 + q] = cons(i, j, k, 2);
    out[nc++ * n3 + q] = 0.5 * (u(i, j, k) + u(i + 1, j, k));
    out[nc++ * n3 + q] = 0.5 * (v(i, j, k) + v(i, j + 1, k));
    out[nc++ * n3 + q] = 0.5 * (w(i, j, k) + w(i, j, k + 1));
}
extern "C" int rh_plot_vars(rh_ctx *c, rh_state *s, int with_scalar, double *out_device)
{
    const rh_box bx = c->bx;
    RH_NEED(s->cons_new, bx); RH_NEED(s->xvel_new, bx_nodal(bx, 0)); RH_NEED(s->yvel_new, bx_nodal(bx, 1)); RH_NEED(s->zvel_new, bx_nodal(bx, 2));
    if (!out_device) return rh_fail("rh_plot_vars: null output");
    k_plot<<<grid3(bx), rh_block(), 0, c->stream>>>(mkB(bx), mkA(s->cons_new), mkA(s->xvel_new), mkA(s->yvel_new), mkA(s->zvel_new), out_device, with_scalar);
    RH_LAUNCH_CHECK(); return 0;
}
