// REMORA_hip_members.cpp -- the member functions of class REMORA whose bodies are replaced by calls into libremora_hip.so.
//
// STATUS: REMORA-side source, NOT compiled in this environment (AMReX absent; see REMORA_hip_glue.H).  Each function keeps the
// reference's signature (Source/REMORA.H:357-648) so that its callers do not change; the file replaces, in Source/TimeIntegration/:
//   REMORA_Advance.cpp, REMORA_setup_step.cpp, REMORA_advance_2d_onestep.cpp, REMORA_advance_2d.cpp, REMORA_advance_3d_ml.cpp
//   (driver level) and REMORA_vert_visc_3d.cpp, REMORA_rhs_t_3d.cpp, REMORA_rhs_uv_3d.cpp, REMORA_uv3dmix.cpp,
//   REMORA_update_massflux_3d.cpp, REMORA_vert_mean_3d.cpp, REMORA_rhs_uv_2d.cpp, REMORA_coriolis.cpp, REMORA_t3dmix.cpp (leaf level:
//   still called by REMORA_prestep.cpp / REMORA_advance_3d.cpp if a maintainer keeps those drivers).
#include <REMORA.H>
#include "REMORA_hip_glue.H"

using namespace amrex;

// ---------------------------------------------------------------------------------------------------------------- context
// rh_params from geom[lev], solverChoice, fixed_dt / fixed_ndtfast_ratio and this rank's box
void REMORA::rh_make_context (int lev)
{
    AMREX_ALWAYS_ASSERT_WITH_MESSAGE(lev == 0, "libremora_hip serves level 0 (max_level = 0)");
    const Box& dom = geom[lev].Domain();
    rh_params p{};
    for (int d = 0; d < 3; ++d) {
        p.domain_lo[d] = dom.smallEnd(d); p.domain_hi[d] = dom.bigEnd(d);
        p.prob_lo[d] = geom[lev].ProbLo(d); p.prob_hi[d] = geom[lev].ProbHi(d);
    }
    p.periodic[0] = geom[lev].isPeriodic(0); p.periodic[1] = geom[lev].isPeriodic(1);
    // one box per rank, laid out as a tensor-product i-j process grid: rank = ry * PX + rx
    AMREX_ALWAYS_ASSERT_WITH_MESSAGE(cons_new[lev]->local_size() == 1, "one box per rank");
    Box bx;
    for (MFIter mfi(*cons_new[lev]); mfi.isValid(); ++mfi) bx = mfi.validbox();
    for (int d = 0; d < 3; ++d) { p.box_lo[d] = bx.smallEnd(d); p.box_hi[d] = bx.bigEnd(d); }
    for (int d = 0; d < 2; ++d) {
        const int len = bx.length(d), n = dom.length(d);
        AMREX_ALWAYS_ASSERT_WITH_MESSAGE(n % len == 0, "boxes must tile the domain evenly (amr.max_grid_size = nx/PX ny/PY nz)");
        p.nranks[d] = n / len;
        p.rank_xy[d] = (bx.smallEnd(d) - dom.smallEnd(d)) / len;
    }
    AMREX_ALWAYS_ASSERT(ParallelDescriptor::MyProc() == p.rank_xy[1] * p.nranks[0] + p.rank_xy[0]);   // DistributionMapping must follow the grid order
    p.dt = fixed_dt; p.ndtfast = fixed_ndtfast_ratio;
    const SolverChoice& sc = solverChoice;
    p.flat_bathymetry = sc.flat_bathymetry; p.tracer_c4 = (sc.tracer_Hadv_scheme == AdvectionScheme::centered4);
    p.uv_c2 = (sc.uv_Hadv_scheme == AdvectionScheme::centered2);
    p.use_coriolis = sc.use_coriolis; p.use_prestep = sc.use_prestep; p.use_uv3dmix = sc.use_uv3dmix;
    p.use_barotropic = sc.use_barotropic; p.use_salt = sc.use_salt;
    p.g = sc.g; p.rho0 = sc.rho0; p.R0 = sc.R0; p.T0 = sc.T0; p.S0 = sc.S0; p.Tcoef = sc.Tcoef; p.Scoef = sc.Scoef;
    p.theta_s = sc.theta_s; p.theta_b = sc.theta_b; p.tcline = sc.tcline;
    // The analytic hooks init_custom_vmix / init_custom_smflux stay REMORA's own (Exec/<case>/prob.cpp): the caller fills Akv / Akt and
    // sustr / svstr, and tells the library not to (RH_VMIX_* / RH_SMFLUX_* only reproduce the regression problems for the tests).
    p.smflux_mode = RH_SMFLUX_FIXED;
    if (sc.vert_mixing_type == VertMixingType::GLS) {
        p.vmix_mode = RH_VMIX_GLS;
        rh_gls_params& G = p.gls;
        G.stability = (sc.gls_stability_type == GLS_StabilityType::Canuto_A) ? RH_GLS_CANUTO_A :
                      (sc.gls_stability_type == GLS_StabilityType::Canuto_B) ? RH_GLS_CANUTO_B : RH_GLS_GALPERIN;
        G.gls_p = sc.gls_p; G.gls_m = sc.gls_m; G.gls_n = sc.gls_n; G.gls_Kmin = sc.gls_Kmin; G.gls_Pmin = sc.gls_Pmin; G.gls_cmu0 = sc.gls_cmu0;
        G.gls_c1 = sc.gls_c1; G.gls_c2 = sc.gls_c2; G.gls_c3m = sc.gls_c3m; G.gls_c3p = sc.gls_c3p; G.gls_sigk = sc.gls_sigk; G.gls_sigp = sc.gls_sigp;
        G.Akk_bak = sc.Akk_bak; G.Akp_bak = sc.Akp_bak; G.Akv_bak = sc.Akv_bak; G.Akt_bak = sc.Akt_bak; G.Zos = sc.Zos; G.Zob = sc.Zob;
    } else {
        p.vmix_mode = RH_VMIX_CONST;          // see above: with REMORA's own set_analytical_vmix kept, Akv / Akt arrive filled (*)
    }
    p.masks_all_one = 0;                      // REMORA may carry land masks (NetCDF grids): keep the final cons *= mskr pass
    if (m_rh.size() <= static_cast<std::size_t>(lev)) m_rh.resize(lev + 1, nullptr);
    RH_CALL(rh_create(&p, (void*)Gpu::gpuStream(), &m_rh[lev]));
    // REMORA::init_bcs has run (REMORA.cpp: constructor -> init_bcs): hand its phys_bc_type over, one variable and lateral face at a
    // time.  RH_PBC_* follows enum struct REMORA_BC (REMORA_IndexDefines.H:53-56) and the variable index follows BCVars, so both
    // casts are the identity; the library redoes the translation of init_bcs.cpp:201-679 itself (rh_get_bc reads it back: a
    // maintainer can assert it equals domain_bcs_type).  Inflow / no-slip values travel as m_bc_extdir_vals.
    for (int v = 0; v <= BCVars::tke_bc; ++v) {
        for (OrientationIter oit; oit; ++oit) {
            const Orientation ori = oit();
            if (ori.coordDir() == 2) continue;                                 // z faces are SlipWall by construction (init_bcs.cpp:130)
            RH_CALL(rh_set_bc(m_rh[lev], v, static_cast<int>(ori), static_cast<int>(phys_bc_type[v][ori]), m_bc_extdir_vals[v][ori]));
        }
    }
    if (ParallelDescriptor::NProcs() > 1)
        RH_CALL(rh_comm_set_exchange(m_rh[lev], rh_mpi_exchange, nullptr, ParallelDescriptor::MyProc(), ParallelDescriptor::NProcs()));
    nfast = rh_nfast(m_rh[lev]);                                               // REMORA::set_weights, REMORA_set_weights.cpp:10-171
    vec_weight1.resize(2 * fixed_ndtfast_ratio + 1); vec_weight2.resize(2 * fixed_ndtfast_ratio + 1);
    RH_CALL(rh_weights(m_rh[lev], vec_weight1.data(), vec_weight2.data()));
}
// (*) a maintainer who keeps set_analytical_vmix (Source/REMORA.cpp:565-573) calls it before rh_setup_step and passes Akv_const = the
//     value it would write only when the profile really is constant; rh_setup_step's own vmix pass is the library's stand-in for it.

void REMORA::rh_free_context (int lev)
{
    if (static_cast<std::size_t>(lev) < m_rh.size() && m_rh[lev]) { RH_CALL(rh_destroy(m_rh[lev])); m_rh[lev] = nullptr; }
}

// one rh_array4 per MultiFab the drivers touch (Source/REMORA.H:189-354 -> include/remora_hip.h: rh_state)
rh_state REMORA::rh_bind_state (int lev)
{
    rh_state s{};
    s.cons_old = rh_view(*cons_old[lev]); s.cons_new = rh_view(*cons_new[lev]);
    s.xvel_old = rh_view(*xvel_old[lev]); s.xvel_new = rh_view(*xvel_new[lev]);
    s.yvel_old = rh_view(*yvel_old[lev]); s.yvel_new = rh_view(*yvel_new[lev]);
    s.zvel_old = rh_view(*zvel_old[lev]); s.zvel_new = rh_view(*zvel_new[lev]);
    s.Zt_avg1 = rh_view(*vec_Zt_avg1[lev]); s.h = rh_view(*vec_hOfTheConfusingName[lev]);
    s.ubar = rh_view(*vec_ubar[lev]); s.vbar = rh_view(*vec_vbar[lev]);
    s.ru = rh_view(*vec_ru[lev]); s.rv = rh_view(*vec_rv[lev]); s.ru2d = rh_view(*vec_ru2d[lev]); s.rv2d = rh_view(*vec_rv2d[lev]);
    s.mskr = rh_view(*vec_mskr[lev]); s.msku = rh_view(*vec_msku[lev]); s.mskv = rh_view(*vec_mskv[lev]); s.mskp = rh_view(*vec_mskp[lev]);
    s.z_w = rh_view(*vec_z_w[lev]); s.z_r = rh_view(*vec_z_r[lev]); s.Hz = rh_view(*vec_Hz[lev]);
    s.Huon = rh_view(*vec_Huon[lev]); s.Hvom = rh_view(*vec_Hvom[lev]);
    s.Akv = rh_view(*vec_Akv[lev]); s.Akt = rh_view(*vec_Akt[lev]);
    s.visc2_p = rh_view(*vec_visc2_p[lev]); s.visc2_r = rh_view(*vec_visc2_r[lev]); s.diff2 = rh_view(*vec_diff2[lev]);
    s.rufrc = rh_view(*vec_rufrc[lev]); s.rvfrc = rh_view(*vec_rvfrc[lev]);
    s.sustr = rh_view(*vec_sustr[lev]); s.svstr = rh_view(*vec_svstr[lev]); s.rdrag = rh_view(*vec_rdrag[lev]);
    s.bustr = rh_view(*vec_bustr[lev]); s.bvstr = rh_view(*vec_bvstr[lev]);
    s.DU_avg1 = rh_view(*vec_DU_avg1[lev]); s.DU_avg2 = rh_view(*vec_DU_avg2[lev]);
    s.DV_avg1 = rh_view(*vec_DV_avg1[lev]); s.DV_avg2 = rh_view(*vec_DV_avg2[lev]);
    s.rubar = rh_view(*vec_rubar[lev]); s.rvbar = rh_view(*vec_rvbar[lev]); s.rzeta = rh_view(*vec_rzeta[lev]);
    s.zeta = rh_view(*vec_zeta[lev]); s.pm = rh_view(*vec_pm[lev]); s.pn = rh_view(*vec_pn[lev]); s.fcor = rh_view(*vec_fcor[lev]);
    s.sstore = rh_view(*vec_sstore[lev]); s.rhoS = rh_view(*vec_rhoS[lev]); s.rhoA = rh_view(*vec_rhoA[lev]); s.bvf = rh_view(*vec_bvf[lev]);
    if (solverChoice.vert_mixing_type == VertMixingType::GLS) {
        s.tke = rh_view(*vec_tke[lev]); s.gls = rh_view(*vec_gls[lev]); s.Lscale = rh_view(*vec_Lscale[lev]);
        s.Akk = rh_view(*vec_Akk[lev]); s.Akp = rh_view(*vec_Akp[lev]);
    } else {
        s.tke = s.gls = s.Lscale = s.Akk = s.Akp = rh_null_view();
    }
    return s;
}

// ---------------------------------------------------------------------------------------------------------------- drivers
// REMORA_Advance.cpp:7-35.  timeStep has already swapped the old / new MultiFab pointers (REMORA_TimeStep.cpp:58-61).
void REMORA::Advance (int lev, Real time, Real /*dt_lev*/, int /*iteration*/, int /*ncycle*/)
{
    BL_PROFILE("REMORA::Advance()");
    rh_state s = rh_bind_state(lev);
    set_smflux(lev, time);                                                     // REMORA's own hook: sustr / svstr (RH_SMFLUX_FIXED)
    RH_CALL(rh_advance(m_rh[lev], &s, istep[lev], time));
}
// REMORA_setup_step.cpp:7-369 (with REMORA::prestep, REMORA_prestep.cpp:36-185)
void REMORA::setup_step (int lev, Real time, Real /*dt_lev*/)
{
    BL_PROFILE("REMORA::setup_step()");
    rh_state s = rh_bind_state(lev);
    RH_CALL(rh_setup_step(m_rh[lev], &s, istep[lev], time));
}
// REMORA_advance_2d_onestep.cpp:6-53
void REMORA::advance_2d_onestep (int lev, Real /*dt_lev*/, Real /*dtfast_lev*/, int my_iif, int /*nfast_counter*/)
{
    rh_state s = rh_bind_state(lev);
    RH_CALL(rh_advance_2d_onestep(m_rh[lev], &s, istep[lev], my_iif));
}
// REMORA_advance_2d.cpp:41-757: the MultiFab arguments are the class's own members (REMORA_advance_2d_onestep.cpp:13-51 passes them),
// so the state is bound from the members as everywhere else
void REMORA::advance_2d (int lev, MultiFab const*, MultiFab const*, MultiFab*, MultiFab*, MultiFab*, MultiFab*, MultiFab*,
                         std::unique_ptr<MultiFab>&, std::unique_ptr<MultiFab>&, std::unique_ptr<MultiFab>&, std::unique_ptr<MultiFab>&,
                         std::unique_ptr<MultiFab>&, std::unique_ptr<MultiFab>&, std::unique_ptr<MultiFab>&,
                         std::unique_ptr<MultiFab>&, std::unique_ptr<MultiFab>&, MultiFab*, MultiFab const*, MultiFab const*, MultiFab const*,
                         MultiFab const*, MultiFab const*, MultiFab const*, MultiFab const*, MultiFab const*, MultiFab const*, MultiFab const*,
                         const Real dtfast_lev, const bool predictor_2d_step, const bool first_2d_step, int my_iif, int& next_indx1)
{
    rh_state s = rh_bind_state(lev);
    RH_CALL(rh_advance_2d(m_rh[lev], &s, istep[lev], dtfast_lev, predictor_2d_step ? 1 : 0, first_2d_step ? 1 : 0, my_iif, &next_indx1));
}
// REMORA_advance_3d_ml.cpp:6-71 + REMORA_advance_3d.cpp:10-387
void REMORA::advance_3d_ml (int lev, Real /*dt_lev*/)
{
    rh_state s = rh_bind_state(lev);
    RH_CALL(rh_advance_3d_ml(m_rh[lev], &s, istep[lev]));
}

// Initialization/REMORA_make_new_level.cpp:617-655.  rh_update_mskp also refills the periodic ghost cells of the four masks, which
// init_masks_from_netcdf does right after this call (REMORA_init_from_netcdf.cpp:408-412): those FillBoundary calls become no-ops.
void REMORA::update_mskp (int lev)
{
    rh_state s = rh_bind_state(lev);
    RH_CALL(rh_update_mskp(m_rh[lev], &s));
}

// ---------------------------------------------------------------------------------------------------------------- leaves
// REMORA_vert_visc_3d.cpp:10-107: the six scratch arrays (Hzk, AK, BC, DC, FC, CF) are not needed, the solve lives in registers / LDS
void REMORA::vert_visc_3d (const Box& phi_bx, const int ioff, const int joff,
                           const Array4<Real      >& phi, const Array4<Real const>& Hz,
                           const Array4<Real      >& /*Hzk*/, const Array4<Real      >& /*AK*/, const Array4<Real      >& Akv,
                           const Array4<Real      >& /*BC*/, const Array4<Real      >& /*DC*/, const Array4<Real      >& /*FC*/,
                           const Array4<Real      >& /*CF*/, const int nnew, const int N, const Real dt_lev)
{
    RH_CALL(rh_vert_visc_3d(m_rh[0], rh_bx(phi_bx), ioff, joff, rh_a4(Array4<Real>(phi, nnew)), rh_a4(Hz), rh_a4(Akv), N, dt_lev));
}
// REMORA_rhs_t_3d.cpp:27-335 (FC scratch not needed)
void REMORA::rhs_t_3d (const Box& bx, const Box& /*gbx*/, const Array4<Real      >& t, const Array4<Real const>& tempstore,
                       const Array4<Real const>& Huon, const Array4<Real const>& Hvom, const Array4<Real const>& Hz,
                       const Array4<Real const>& pn, const Array4<Real const>& pm, const Array4<Real const>& W,
                       const Array4<Real      >& /*FC*/, const Array4<Real const>& msku, const Array4<Real const>& mskv,
                       int nrhs, int nnew, int N, const Real dt_lev)
{
    RH_CALL(rh_rhs_t_3d(m_rh[0], rh_bx(bx), rh_a4(Array4<Real>(t, nnew)), rh_a4(Array4<Real const>(tempstore, nrhs)), rh_a4(Huon), rh_a4(Hvom),
                        rh_a4(Hz), rh_a4(pn), rh_a4(pm), rh_a4(W), rh_a4(msku), rh_a4(mskv), N, dt_lev));
}
// REMORA_coriolis.cpp:10-40
void REMORA::coriolis (const Box& xbx, const Box& ybx, const Array4<Real const>& uold, const Array4<Real const>& vold,
                       const Array4<Real      >& ru, const Array4<Real      >& rv, const Array4<Real const>& Hz,
                       const Array4<Real const>& fomn, int nrhs, int nr)
{
    RH_CALL(rh_coriolis(m_rh[0], rh_bx(xbx), rh_bx(ybx), rh_a4(uold), rh_a4(vold), rh_a4(ru), rh_a4(rv), rh_a4(Hz), rh_a4(fomn), nrhs, nr));
}
// REMORA_rhs_uv_3d.cpp:31-385.  The library's entry point folds the Coriolis term in when asked to; REMORA calls coriolis() itself
// first (REMORA_setup_step.cpp:296-300), so use_coriolis = 0 here.  Hz and fcor are then not read.
void REMORA::rhs_uv_3d (const Box& xbx, const Box& ybx, const Array4<Real const>& uold, const Array4<Real const>& vold,
                        const Array4<Real      >& ru, const Array4<Real      >& rv, const Array4<Real      >& rufrc, const Array4<Real      >& rvfrc,
                        const Array4<Real const>& sustr, const Array4<Real const>& svstr, const Array4<Real const>& bustr, const Array4<Real const>& bvstr,
                        const Array4<Real const>& Huon, const Array4<Real const>& Hvom, const Array4<Real const>& pm, const Array4<Real const>& pn,
                        const Array4<Real const>& W, const Array4<Real      >& /*FC*/, int nrhs, int N)
{
    rh_state s = rh_bind_state(0);
    RH_CALL(rh_rhs_uv_3d(m_rh[0], rh_bx(xbx), rh_bx(ybx), rh_a4(uold), rh_a4(vold), rh_a4(ru), rh_a4(rv), rh_a4(rufrc), rh_a4(rvfrc), rh_a4(sustr),
                         rh_a4(svstr), rh_a4(bustr), rh_a4(bvstr), rh_a4(Huon), rh_a4(Hvom), rh_a4(pm), rh_a4(pn), rh_a4(W), s.Hz, s.fcor, 0, nrhs, N));
}
// REMORA_uv3dmix.cpp:6-149 (3-D call; the 2-D call of REMORA_advance_2d.cpp:536-538 is inside rh_advance_2d)
void REMORA::uv3dmix (const Box& xbx, const Box& ybx, const Array4<Real      >& u, const Array4<Real      >& v,
                      const Array4<Real const>& uold, const Array4<Real const>& vold, const Array4<Real      >& rufrc, const Array4<Real      >& rvfrc,
                      const Array4<Real const>& visc2_p, const Array4<Real const>& visc2_r, const Array4<Real const>& Hz,
                      const Array4<Real const>& pm, const Array4<Real const>& pn, const Array4<Real const>& mskp, int nrhs, int nnew, const Real dt_lev)
{
    RH_CALL(rh_uv3dmix(m_rh[0], rh_bx(xbx), rh_bx(ybx), rh_a4(u), rh_a4(v), rh_a4(uold), rh_a4(vold), rh_a4(rufrc), rh_a4(rvfrc), rh_a4(visc2_p),
                       rh_a4(visc2_r), rh_a4(Hz), rh_a4(pm), rh_a4(pn), rh_a4(mskp), nrhs, nnew, dt_lev));
}
// REMORA_t3dmix.cpp:6-58
void REMORA::t3dmix (const Box& bx, const Array4<Real      >& state, const Array4<Real      >& state_rhs, const Array4<Real const>& diff2,
                     const Array4<Real const>& Hz, const Array4<Real const>& pm, const Array4<Real const>& pn,
                     const Array4<Real const>& msku, const Array4<Real const>& mskv, const Real dt_lev, const int ncomp)
{
    RH_CALL(rh_t3dmix(m_rh[0], rh_bx(bx), rh_a4(state), rh_a4(state_rhs), rh_a4(diff2), rh_a4(Hz), rh_a4(pm), rh_a4(pn), rh_a4(msku), rh_a4(mskv), dt_lev, ncomp));
}
// REMORA_vert_mean_3d.cpp:10-47 (DC, CF scratch not needed)
void REMORA::vert_mean_3d (const Box& phi_bx, const int ioff, const int joff, const Array4<Real      >& phi, const Array4<Real const>& Hz,
                           const Array4<Real const>& Dphi_avg1, const Array4<Real      >& /*DC*/, const Array4<Real      >& /*CF*/,
                           const Array4<Real const>& dxlen, const Array4<Real const>& msk, const int nnew, const int N)
{
    RH_CALL(rh_vert_mean_3d(m_rh[0], rh_bx(phi_bx), ioff, joff, rh_a4(Array4<Real>(phi, nnew)), rh_a4(Hz), rh_a4(Dphi_avg1), rh_a4(dxlen), rh_a4(msk), N));
}
// REMORA_update_massflux_3d.cpp:24-122 (DC, FC scratch not needed)
void REMORA::update_massflux_3d (const Box& bx, const int ioff, const int joff, const Array4<Real      >& phi, const Array4<Real      >& phibar,
                                 const Array4<Real      >& Hphi, const Array4<Real const>& Hz, const Array4<Real const>& pm_or_pn,
                                 const Array4<Real const>& Dphi_avg1, const Array4<Real const>& Dphi_avg2, const Array4<Real      >& /*DC*/,
                                 const Array4<Real      >& /*FC*/, const Array4<Real const>& msk, const int nnew)
{
    const int N = Geom(0).Domain().size()[2] - 1;
    RH_CALL(rh_update_massflux_3d(m_rh[0], rh_bx(bx), ioff, joff, rh_a4(Array4<Real>(phi, nnew)), rh_a4(phibar), rh_a4(Hphi), rh_a4(Hz), rh_a4(pm_or_pn),
                                  rh_a4(Dphi_avg1), rh_a4(Dphi_avg2), rh_a4(msk), N));
}
// REMORA_rhs_uv_2d.cpp:20-238
void REMORA::rhs_uv_2d (const Box& xbx, const Box& ybx, const Array4<Real const>& ubar, const Array4<Real const>& vbar,
                        const Array4<Real      >& rhs_ubar, const Array4<Real      >& rhs_vbar, const Array4<Real const>& DUon, const Array4<Real const>& DVom,
                        const int krhs)
{
    RH_CALL(rh_rhs_uv_2d(m_rh[0], rh_bx(xbx), rh_bx(ybx), rh_a4(ubar), rh_a4(vbar), rh_a4(rhs_ubar), rh_a4(rhs_vbar), rh_a4(DUon), rh_a4(DVom), krhs));
}

// ---------------------------------------------------------------------------------------------------------------- ghost cells
// REMORA::FillPatch at level 0 (BoundaryConditions/REMORA_FillPatch.cpp:16-155) for the arrays the drivers above do not fill themselves:
//   RH_CALL(rh_fill_patch(m_rh[0], rh_view(mf), ng, rh_view(*vec_mskr[0]), bccomp, icomp, ncomp));
// with ng = {mf.nGrowVect()[0], mf.nGrowVect()[1], 0}; FillBoundary alone is rh_fill_boundary.  Across ranks the exchange goes through
// rh_mpi_exchange (REMORA_hip_glue.H) or the built-in RCCL communicator (rh_comm_unique_id on rank 0, ParallelDescriptor::Bcast of the
// 128 bytes, rh_comm_init on every rank).
