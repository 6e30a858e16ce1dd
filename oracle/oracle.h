/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of REMORA's TimeIntegration hot path, pass by pass,
 * in the same arithmetic order as the reference's amrex::ParallelFor lambdas.
 * Pinned end-to-end against the reference's own gold plotfiles
 * (Tests/REMORA_Gold_Files/{Upwelling,Seamount,DoubleGyre,Advection}, 1e-11).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * All "file:line" citations are relative to the reference tree.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "fab.h"

#define NCONS 3
#define NGROW 2

/* mathematical BC types, Source/REMORA_IndexDefines.H:61-73 */
enum { BC_BOGUS=-666, BC_REFLECT_ODD=-1, BC_INT_DIR=0, BC_REFLECT_EVEN=1, BC_FOEXTRAP=2, BC_EXT_DIR=3,
       BC_CLAMPED=4, BC_CHAPMAN=5, BC_FLATHER=6, BC_ORLANSKI_RAD=7, BC_ORLANSKI_RAD_NUDGE=8 };
/* physical boundary kinds, enum struct REMORA_BC, Source/REMORA_IndexDefines.H:53-56 */
enum { PBC_SYMMETRY=0, PBC_INFLOW, PBC_OUTFLOW, PBC_NO_SLIP_WALL, PBC_SLIP_WALL, PBC_PERIODIC, PBC_CLAMPED, PBC_CHAPMAN,
       PBC_FLATHER, PBC_ORLANSKI_RAD, PBC_ORLANSKI_RAD_NUDGE, PBC_UNDEFINED };
#define BCV_USER 10   /* temp, salt, scalar, u, v, w, ubar, vbar, zeta, tke: the variables of remora.boundary_per_variable */
/* BCVars, Source/REMORA_IndexDefines.H:15-34 */
enum { BCV_CONS=0, BCV_TEMP=0, BCV_SALT, BCV_SCALAR, BCV_XVEL=NCONS, BCV_YVEL, BCV_ZVEL, BCV_UBAR, BCV_VBAR,
       BCV_ZETA, BCV_TKE, BCV_FOEXTRAP_PERIODIC, BCV_FOEXTRAP, BCV_U2D_SIMPLE, BCV_V2D_SIMPLE, BCV_NUM };

typedef struct { int lo[3], hi[3]; } BCRec;

enum { CASE_UPWELLING=0, CASE_SEAMOUNT=1, CASE_DOUBLEGYRE=2, CASE_ADVECTION=3, CASE_DOUBLYPERIODIC=4, CASE_CHANNEL=5 };

/* vertical_mixing_type = gls: SolverChoice's GLS block, Source/REMORA_DataStruct.H:203-268,399-400,427-445
 * stab: 0 Canuto_A, 1 Canuto_B, 2 Galperin */
typedef struct { int on, stab; double p,m,n,Kmin,Pmin,cmu0,c1,c2,c3m,c3p,sigk,sigp,Akk_bak,Akp_bak,Akv_bak,Akt_bak,Zos,Zob; } GlsParams;

typedef struct {
    /* geometry */
    int    ncell[3];
    double prob_lo[3], prob_hi[3];
    int    periodic[2];
    int    slipwall[2];          /* per direction (x,y): 1 = SlipWall on both sides (else periodic) */
    /* time stepping */
    double fixed_dt;
    int    ndtfast;
    /* SolverChoice subset, Source/REMORA_DataStruct.H:351-483 */
    int    flat_bathymetry, tracer_c4, uv_c2, use_coriolis, use_prestep, use_uv3dmix, use_barotropic, use_salt;
    double g, rho0, R0, T0, S0, Tcoef, Scoef, theta_s, theta_b, tcline, rdrag;
    double coriolis_f0, coriolis_beta;
    int    hmix_constant;        /* horizontal_mixing_type = constant */
    double visc2, tnu2[NCONS];
    /* problem */
    int    prob;                 /* CASE_* */
    double u_0, v_0;             /* Advection / DoublyPeriodic prob.u_0 / prob.v_0 */
    double z0, zRef, uRef;       /* DoublyPeriodic prob.z0 / prob.zRef / prob.uRef (log-layer x-velocity) */
    GlsParams gls;
    int    threads;              /* OpenMP threads for the timing baseline (0/1 = serial) */
    /* bc.<var>.type (remora.boundary_per_variable = 1) or bc.<side>.type: PBC_* per variable (BCVars order) and amrex orientation
     * (0 xlo, 1 ylo, 3 xhi, 4 yhi; z is always SlipWall).  bc_set = 0: the table is ignored and every non-periodic side is
     * SlipWall (the pinned cases).  bc_val: bc.<>.velocity[component] / bc.<>.scalar of an inflow or no-slip side. */
    int    bc_set;
    int    bc_kind[BCV_USER][6];
    double bc_val[BCV_USER][6];
} Params;

typedef struct {
    Box  domain;                 /* cell-centred */
    int  per[2];
    const Params *prm;
    BCRec bcs[BCV_NUM];
    double extdir[BCV_NUM][6];
    int  phys[BCV_USER][6];      /* phys_bc_type */
} Geom;

/* ---- leaves (leaves.c) ---- */
void o_rho_eos(const Geom*,Box bx,Fab state,Fab rho,Fab rhoA,Fab rhoS,Fab bvf,Fab Hz,Fab z_w,Fab z_r,Fab h,Fab mskr,int N);
void o_prsgrd(const Geom*,Box phi_bx,Box phi_gbx,Box utbx,Box vtbx,Fab ru,Fab rv,Fab pn,Fab pm,Fab rho,Fab FC,
              Fab Hz,Fab z_r,Fab z_w,Fab msku,Fab mskv,int nrhs,int N);
void o_prestep_t_advection(const Geom*,Box tbx,Box gbx,Fab tempold,Fab tempcache,Fab Hz,Fab Huon,Fab Hvom,Fab W,Fab DC,Fab FC,
              Fab tempstore,Fab z_w,Fab h,Fab pm,Fab pn,Fab msku,Fab mskv,int iic,int ntfirst,int nrhs,int N,double dt);
void o_prestep_diffusion(const Geom*,Box vel_bx,Box gbx,int ioff,int joff,Fab vel,Fab vel_old,Fab rvel,Fab Hz,Fab Akv,
              Fab DC,Fab FC,Fab sstr,Fab bstr,Fab z_r,Fab pm,Fab pn,int iic,int ntfirst,int nnew,int nstp,int nrhs,int N,
              double lambda,double dt);
void o_t3dmix(const Geom*,Box bx,Fab state,Fab state_rhs,Fab diff2,Fab Hz,Fab pm,Fab pn,Fab msku,Fab mskv,double dt,int ncomp);
void o_coriolis(Box xbx,Box ybx,Fab uold,Fab vold,Fab ru,Fab rv,Fab Hz,Fab fomn,int nrhs,int nr);
void o_rhs_uv_3d(const Geom*,Box xbx,Box ybx,Fab uold,Fab vold,Fab ru,Fab rv,Fab rufrc,Fab rvfrc,Fab sustr,Fab svstr,
              Fab bustr,Fab bvstr,Fab Huon,Fab Hvom,Fab pm,Fab pn,Fab W,Fab FC,int nrhs,int N);
void o_uv3dmix(Box xbx,Box ybx,Fab u,Fab v,Fab uold,Fab vold,Fab rufrc,Fab rvfrc,Fab visc2_p,Fab visc2_r,Fab Hz,
              Fab pm,Fab pn,Fab mskp,int nrhs,int nnew,double dt);
void o_vert_visc_3d(Box phi_bx,int ioff,int joff,Fab phi,Fab Hz,Fab Hzk,Fab AK,Fab Akv,Fab BC,Fab DC,Fab FC,Fab CF,
              int nnew,int N,double dt);
void o_vert_mean_3d(Box phi_bx,int ioff,int joff,Fab phi,Fab Hz,Fab Dphi_avg1,Fab DC,Fab CF,Fab dxlen,Fab msk,int nnew,int N);
void o_update_massflux_3d(const Geom*,Box bx,int ioff,int joff,Fab phi,Fab phibar,Fab Hphi,Fab Hz,Fab pm_or_pn,
              Fab Dphi_avg1,Fab Dphi_avg2,Fab DC,Fab FC,Fab msk,int nnew);
void o_rhs_t_3d(const Geom*,Box bx,Box gbx,Fab t,Fab sstore,Fab Huon,Fab Hvom,Fab Hz,Fab pn,Fab pm,Fab W,Fab FC,
              Fab msku,Fab mskv,int nrhs,int nnew,int N,double dt);
void o_rhs_uv_2d(const Geom*,Box xbx,Box ybx,Fab ubar,Fab vbar,Fab rhs_ubar,Fab rhs_vbar,Fab DUon,Fab DVom,int krhs);
void o_omega(Box gbx1,Fab W,Fab Huon,Fab Hvom,Fab z_w,Fab h,int N);
/* returns nfast; w1,w2 must hold 2*ndtfast+1 doubles */
int  o_set_weights(int ndtfast,double *w1,double *w2);
/* stretching curves Cs/sc at one k (k-only part of stretch_transform) */
void o_stretch_coeffs(int k,int nz,double theta_s,double theta_b,double *sc_w,double *Cs_w,double *sc_r,double *Cs_r);

/* ---- ghost fill / physical BCs (bc.c) ---- */
/* one-box "MultiFab": data on the grown box, plus the valid box and index type */
typedef struct { Fab f; Box valid; int ix[3]; int ng[3]; } MF;
MF   mf_alloc(Box cellbox,int ix0,int ix1,int ix2,int nc,int ng0,int ng1,int ng2);
void mf_free(MF*);
void o_init_bcs(Geom*);
void o_fill_boundary(const Geom*,MF *mf,int icomp,int ncomp);     /* single-box periodic FillBoundary */
void o_physbc(const Geom*,MF *mf,Fab msk,int icomp,int ncomp,int bccomp,int n_not_fill);
/* REMORA::FillPatch at level 0 == FillBoundary + physbc (BoundaryConditions/REMORA_FillPatch.cpp:96-125) */
void o_fillpatch(const Geom*,MF *mf,Fab msk,int bccomp,int icomp,int ncomp);
/* the same with the arrays the radiation conditions read: mf_calc (a Fab with p = NULL stands for the reference's empty MultiFab)
 * and, for cell-centred arrays, msku / mskv (REMORA_PhysBCFunct.cpp:21-23, REMORA_FillPatch.cpp:128-129) */
void o_physbc_calc(const Geom*,MF *mf,Fab msk,int icomp,int ncomp,int bccomp,int n_not_fill,Fab calc,Fab msku,Fab mskv);
void o_fillpatch_calc(const Geom*,MF *mf,Fab msk,int bccomp,int icomp,int ncomp,Fab calc,Fab msku,Fab mskv);
/* mathematical types of one BCVars entry: out6 = lo(0..2), hi(0..2) */
void o_get_bc(const Geom*,int bccomp,int *out6);

/* ---- GLS closure (gls.c) ---- */
void o_gls_prestep(const Geom*,Box vbx,Fab gls,Fab tke,Fab W,Fab Huon,Fab Hvom,Fab Hz,Fab pm,Fab pn,Fab msku,Fab mskv,
                   int nstp,int nnew,int iic,int ntfirst,int N,double dt_lev);
void o_gls_corrector(const Geom*,Box vbx,Fab gls,Fab tke,Fab W,Fab Akv,Fab Akt,Fab Akk,Fab Akp,Fab Lscale,Fab mskr,Fab msku,Fab mskv,
                     Fab Hz,Fab u,Fab v,Fab pm,Fab pn,Fab Huon,Fab Hvom,Fab z_w,Fab sustr,Fab svstr,Fab bustr,Fab bvstr,Fab bvf,
                     int nstp,int nnew,int N,double dt_lev);

/* ---- driver (driver.c) ---- */
typedef struct OState OState;
OState* o_create(const Params*);
void    o_destroy(OState*);
void    o_advance(OState*);                       /* one baroclinic step (timeStep + Advance) */
void    o_setup_step(OState*);                    /* exposed for phase-by-phase tests */
void    o_advance_2d_onestep(OState*,int my_iif);
void    o_advance_3d_ml(OState*);
void    o_begin_step(OState*);                    /* swap old/new, time bookkeeping */
void    o_end_step(OState*);
void    o_update_mskp(OState*);                   /* mskp from mskr, make_new_level.cpp:617-655 */
int     o_nfast(const OState*);
int     o_istep(const OState*);
/* field access by name: returns pointer + box + ncomp (0 on unknown name) */
int     o_field(OState*,const char *name,double **p,int lo[3],int hi[3],int *nc);
/* plotfile variables (temp,salt,[scalar],x_velocity,y_velocity,z_velocity), FAB order, valid box */
void    o_plot(OState*,int with_scalar,double *out);
double  o_est_timestep(OState*,double cfl);
void    o_sum_iq(OState*,double *out4);

#endif
