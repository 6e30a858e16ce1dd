/*
 * oracle/driver.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Single-level, single-box restatement of the REMORA drivers that surround the leaf routines:
 *   init            Source/REMORA.cpp:246-342,608-702, Initialization/REMORA_make_new_level.cpp:234-614,
 *                   Initialization/REMORA_init.cpp:11-154, Exec/{Upwelling,Seamount,DoubleGyre,Advection}/prob.cpp
 *   stretch         Source/Utils/REMORA_DepthStretchTransform.H:12-249
 *   time loop       TimeIntegration/REMORA_TimeStep.cpp:48-66, REMORA_Advance.cpp:7-35
 *   setup_step      TimeIntegration/REMORA_setup_step.cpp:7-369, REMORA_prestep.cpp:36-185
 *   barotropic      TimeIntegration/REMORA_advance_2d_onestep.cpp:6-53, REMORA_advance_2d.cpp:41-757
 *   baroclinic      TimeIntegration/REMORA_advance_3d_ml.cpp:6-71, REMORA_advance_3d.cpp:10-387
 *   plot variables  IO/REMORA_Plotfile.cpp:141-235 (cell-centred averages of face velocities)
 */
#define ORACLE_SERIAL_LOOPS
#include <stdio.h>
#include "oracle.h"

struct OState {
    Params prm; Geom geom;
    int nx,ny,nz,N;
    Box bx;                      /* valid cell box == domain */
    double dt, t_old, t_new; int istep;
    int nfast; double *weight1,*weight2;
    double dxi[2];
    /* state */
    MF cons_old,cons_new,xvel_old,xvel_new,yvel_old,yvel_new,zvel_old,zvel_new;
    MF Zt_avg1,h,ubar,vbar,ru,rv,ru2d,rv2d;
    MF mskr,msku,mskv,mskp;
    MF z_w,z_r,Hz,Huon,Hvom,Akv,Akt,visc2_p,visc2_r,diff2;
    MF rufrc,rvfrc,sustr,svstr,rdrag,bustr,bvstr,DU_avg1,DU_avg2,DV_avg1,DV_avg2,rubar,rvbar,rzeta,zeta;
    MF pm,pn,fcor,sstore,rhoS,rhoA,bvf;
    MF tke,gls,Lscale,Akk,Akp;   /* GLS closure, make_new_level.cpp:499-503 */
};

#define F(m) ((s)->m.f)

static void swap_mf(MF *a,MF *b){ MF t=*a; *a=*b; *b=t; }

/* ------------------------------------------------------------------ stretch_transform
 * Source/Utils/REMORA_DepthStretchTransform.H:12-249 (z_phys_nd / s_r / s_w are not on the hot path and are omitted) */
static void stretch_transform(OState *s)
{
    const Params *P=&s->prm;
    int N=s->nz;                                                               /* here N = nz (:49) */
    Box gbx3=box_grow(s->bx,NGROW+1,NGROW+1,0);
    Box gbx3D=box_slab(gbx3,2,0);
    Box wgbx3=box_nodal(gbx3,2);
    double hc=-dmin(P->prob_hi[2],-P->tcline);                                 /* :51 */
    Fab z_w=F(z_w), z_r=F(z_r), Hz=F(Hz), h=F(h), Zt=F(Zt_avg1);
    FORBOX(wgbx3,i,j,k) V(z_w,i,j,k)=VN(h,i,j,0,0);                            /* :55-58 */
    FORBOX2D(gbx3D,i,j){
        for(int k=0;k<=N;k++){
            double sc_w,Cs_w,sc_r,Cs_r;
            o_stretch_coeffs(k,N,P->theta_s,P->theta_b,&sc_w,&Cs_w,&sc_r,&Cs_r);
            double cff_w=hc*sc_w, cff1_w=Cs_w, cff_r=hc*sc_r, cff1_r=Cs_r;
            double hwater=VN(h,i,j,0,0);
            double hinv=1.0/(hc+hwater);
            double cff2_r=(cff_r+cff1_r*hwater)*hinv;
            double cff2_w=(cff_w+cff1_w*hwater)*hinv;
            if(k==N) V(z_w,i,j,N)=V(Zt,i,j,0);
            else if(k==0){ VN(h,i,j,0,1)=V(Zt,i,j,0)+(V(Zt,i,j,0)+hwater)*cff2_w; V(z_w,i,j,0)=VN(h,i,j,0,1); }
            else V(z_w,i,j,k)=V(Zt,i,j,0)+(V(Zt,i,j,0)+hwater)*cff2_w;
            if(k!=N) V(z_r,i,j,k)=V(Zt,i,j,0)+(V(Zt,i,j,0)+hwater)*cff2_r;
        }
    }
    FORBOX(gbx3,i,j,k) V(Hz,i,j,k)=V(z_w,i,j,k+1)-V(z_w,i,j,k);                /* :192-195 */
    o_fill_boundary(&s->geom,&s->z_w,0,1);                                     /* :198-201 */
    o_fill_boundary(&s->geom,&s->z_r,0,1);
    o_fill_boundary(&s->geom,&s->Hz,0,1);
}

/* ------------------------------------------------------------------ problem hooks (Exec/<case>/prob.cpp) */
static void init_bathymetry(OState *s)
{
    const Params *P=&s->prm; Fab h=F(h);
    double dx0=(P->prob_hi[0]-P->prob_lo[0])/s->nx, dx1=(P->prob_hi[1]-P->prob_lo[1])/s->ny;
    switch(P->prob){
    case CASE_UPWELLING: {                                                     /* Exec/Upwelling/prob.cpp:21-80 */
        fab_setval(h,P->prob_hi[2]);
        int Lm=s->nx, Mm=s->ny;
        Box g2=box_slab(box_grow(s->bx,NGROW,NGROW,0),2,0);
        if(P->periodic[1]){
            FORBOX2D(g2,i,j){
                int iFort=i+1;
                double val1=(iFort<=Lm/2.0)?iFort:Lm+1-iFort; val1-=0.5;
                double adj=dx1/1000.0;
                VN(h,i,j,0,0)=dmin(-P->prob_lo[2],(84.5+66.526*tanh((val1*adj-10.0)/7.0)));
            }
        } else if(P->periodic[0]){
            FORBOX2D(g2,i,j){
                int jFort=j+1;
                double val1=(jFort<=Mm/2.0)?jFort:Mm+1-jFort; val1-=0.5;
                double adj=dx0/1000.0;
                VN(h,i,j,0,0)=dmin(-P->prob_lo[2],(84.5+66.526*tanh((val1*adj-10.0)/7.0)));
            }
        }
    } break;
    case CASE_SEAMOUNT: {                                                      /* Exec/Seamount/prob.cpp:25-72 */
        fab_setval(h,P->prob_hi[2]);
        Box g3=box_slab(box_grow(s->bx,NGROW+1,NGROW+1,0),2,0);
        double Xsize=320000.0, Esize=320000.0, depth=5000.0;
        if(!P->flat_bathymetry){
            FORBOX2D(g3,i,j){
                double x=P->prob_lo[0]+(i+0.5)*dx0, y=P->prob_lo[1]+(j+0.5)*dx1;
                double val1=(x-0.5*Xsize)/40000.0, val2=(y-0.5*Esize)/40000.0;
                VN(h,i,j,0,0)=depth-4500.0*exp(-(val1*val1+val2*val2));
            }
        } else FORBOX2D(g3,i,j){ VN(h,i,j,0,0)=-P->prob_lo[2]; VN(h,i,j,0,1)=VN(h,i,j,0,0); }
    } break;
    case CASE_DOUBLEGYRE: fab_setval(h,500.0); break;                          /* Exec/DoubleGyre/prob.cpp:21-28 */
    case CASE_CHANNEL: fab_setval(h,50.0); break;                              /* Exec/Channel_Test/prob.cpp:20-52 */
    case CASE_DOUBLYPERIODIC: {                                                /* Exec/DoublyPeriodic/prob.cpp:32-105 */
        fab_setval(h,P->prob_hi[2]);
        int Lm=s->nx, Mm=s->ny;
        Box g2=box_slab(box_grow(s->bx,NGROW,NGROW,0),2,0);
        if(!P->flat_bathymetry){
            if(P->periodic[1]){
                FORBOX2D(g2,i,j){
                    int iFort=i+1;
                    double val1=(iFort<=Lm/2.0)?iFort:Lm+1-iFort;
                    VN(h,i,j,0,0)=dmin(-P->prob_lo[2],(84.5+66.526*tanh((val1-10.0)/7.0)));
                }
            } else if(P->periodic[0]){
                FORBOX2D(g2,i,j){
                    int jFort=j+1;
                    double val1=(jFort<=Mm/2.0)?jFort:Mm+1-jFort;
                    VN(h,i,j,0,0)=dmin(-P->prob_lo[2],(84.5+66.526*tanh((val1-10.0)/7.0)));
                }
            }
        } else FORBOX2D(g2,i,j){ VN(h,i,j,0,0)=-P->prob_lo[2]; VN(h,i,j,0,1)=VN(h,i,j,0,0); }
    } break;
    case CASE_ADVECTION: {                                                     /* Exec/Advection/prob.cpp:29-52 */
        fab_setval(h,P->prob_hi[2]);
        Box b=fab_box(h);
        FORBOX2D(b,i,j){ VN(h,i,j,0,0)=-P->prob_lo[2]; VN(h,i,j,0,1)=VN(h,i,j,0,0); }
    } break;
    }
}

static void init_prob(OState *s)
{
    const Params *P=&s->prm;
    Fab st=F(cons_new), z_r=F(z_r);
    double dx0=(P->prob_hi[0]-P->prob_lo[0])/s->nx, dx1=(P->prob_hi[1]-P->prob_lo[1])/s->ny;
    double T0=P->T0,S0=P->S0;
    switch(P->prob){
    case CASE_UPWELLING:                                                       /* Exec/Upwelling/prob.cpp:101-161 */
        FORBOX(s->bx,i,j,k){ double z=V(z_r,i,j,k);
            VN(st,i,j,k,0)=T0+8.0*exp(z/50.0); if(P->use_salt) VN(st,i,j,k,1)=S0; VN(st,i,j,k,2)=0.0; }
        break;
    case CASE_SEAMOUNT:                                                        /* Exec/Seamount/prob.cpp:93-165 */
        FORBOX(s->bx,i,j,k){ double z=V(z_r,i,j,k);
            VN(st,i,j,k,0)=T0+7.5*exp(z/1000.0); if(P->use_salt) VN(st,i,j,k,1)=S0; VN(st,i,j,k,2)=0.0; }
        break;
    case CASE_DOUBLEGYRE: {                                                    /* Exec/DoubleGyre/prob.cpp:49-114 */
        double val1=(44.69/39.382)*(44.69/39.382);
        double val2=val1*(P->rho0*100.0/P->g)*(5.0e-5/((42.689/44.69)*(42.689/44.69)));
        FORBOX(s->bx,i,j,k){
            double y=P->prob_lo[1]+(j+0.5)*dx1, z=V(z_r,i,j,k), yextent=P->prob_hi[1]-P->prob_lo[1];
            double val3=T0+val2*exp(z/100.0)*(10.0-0.4*tanh(z/100.0));
            double val4=y/yextent;
            VN(st,i,j,k,0)=val3-3.0*val4;
            if(P->use_salt) VN(st,i,j,k,1)=34.5-0.001*z-val4;
            VN(st,i,j,k,2)=0.0;
        }
    } break;
    case CASE_DOUBLYPERIODIC:                                                  /* Exec/DoublyPeriodic/prob.cpp:126-180 */
        FORBOX(s->bx,i,j,k){
            double z=V(z_r,i,j,k);
            VN(st,i,j,k,0)=T0+8.0*exp(z/50.0);
            if(P->use_salt) VN(st,i,j,k,1)=S0;
            double xcent=0.5*(P->prob_lo[0]+P->prob_hi[0]), ycent=0.5*(P->prob_lo[1]+P->prob_hi[1]);
            double x=P->prob_lo[0]+(i+0.5)*dx0-xcent, y=P->prob_lo[1]+(j+0.5)*dx1-ycent;
            double r2=x*x+y*y, rad=0.1*(P->prob_hi[0]-P->prob_lo[0]), radsq=rad*rad;
            VN(st,i,j,k,2)=(r2<radsq)?1.0:0.0;
        }
        break;
    case CASE_CHANNEL:                                                         /* Exec/Channel_Test/prob.cpp:73-150 */
        FORBOX(s->bx,i,j,k){
            double z=V(z_r,i,j,k), y=P->prob_lo[1]+(j+0.5)*dx1;
            VN(st,i,j,k,0)=T0+z/(9.8*1.7);
            if(P->use_salt) VN(st,i,j,k,1)=S0+y/(9.8*760.0);
            VN(st,i,j,k,2)=0.0;
        }
        break;
    case CASE_ADVECTION:                                                       /* Exec/Advection/prob.cpp:72-157 */
        FORBOX(s->bx,i,j,k){
            VN(st,i,j,k,0)=T0;
            double xcent=0.5*(P->prob_lo[0]+P->prob_hi[0]), ycent=0.5*(P->prob_lo[1]+P->prob_hi[1]);
            double x=P->prob_lo[0]+(i+0.5)*dx0-xcent, y=P->prob_lo[1]+(j+0.5)*dx1-ycent;
            double r2=x*x+y*y, rad=0.1*(P->prob_hi[0]-P->prob_lo[0]), radsq=rad*rad;
            if(P->use_salt) VN(st,i,j,k,1)=S0;
            VN(st,i,j,k,2)=exp(-r2/(2.*radsq));
        }
        break;
    }
    double u0=(P->prob==CASE_ADVECTION)?P->u_0:0.0, v0=(P->prob==CASE_ADVECTION||P->prob==CASE_DOUBLYPERIODIC)?P->v_0:0.0;
    if(P->prob==CASE_DOUBLYPERIODIC){                                          /* Exec/DoublyPeriodic/prob.cpp:182-198: log layer */
        Box b=s->xvel_new.valid; Fab f=F(xvel_new);
        FORBOX(b,i,j,k){ double z=-V(z_r,i,j,k);
            V(f,i,j,k)=P->u_0+P->uRef*log((z+P->z0)/P->z0)/log((P->zRef+P->z0)/P->z0); }
    } else if(P->prob==CASE_CHANNEL){                                          /* Exec/Channel_Test/prob.cpp:126-141: sheared, wavy along-channel flow */
        Box b=s->xvel_new.valid; Fab f=F(xvel_new);
        FORBOX(b,i,j,k){
            double z=V(z_r,i,j,k), depth=50.0;
            double x=P->prob_lo[0]+(i+0.5)*dx0, y=P->prob_lo[1]+(j+0.5)*dx1;
            double val1=x*6.28318530718*1.0e-5, val2=y*6.28318530718*1.0e-5;
            V(f,i,j,k)=(depth+z)*0.01+0.05*sin(val1)*sin(val2);
        }
    } else
    { Box b=s->xvel_new.valid; Fab f=F(xvel_new); FORBOX(b,i,j,k) V(f,i,j,k)=u0; }
    { Box b=s->yvel_new.valid; Fab f=F(yvel_new); FORBOX(b,i,j,k) V(f,i,j,k)=v0; }
    { Box b=s->zvel_new.valid; Fab f=F(zvel_new); FORBOX(b,i,j,k) V(f,i,j,k)=0.0; }
}

static void init_vmix(OState *s)
{
    const Params *P=&s->prm; Fab Akv=F(Akv), Akt=F(Akt), z_w=F(z_w);
    Box b=box_grow(s->Akv.valid,NGROW,NGROW,0);
    FORBOX(b,i,j,k){
        switch(P->prob){
        case CASE_UPWELLING: case CASE_ADVECTION: case CASE_DOUBLYPERIODIC: case CASE_CHANNEL:   /* Exec/Upwelling/prob.cpp:163-185, DoublyPeriodic/prob.cpp:222-244, Channel_Test/prob.cpp:152-176 */
            V(Akv,i,j,k)=2.0e-03+8.0e-03*exp(V(z_w,i,j,k)/150.0);
            VN(Akt,i,j,k,0)=1.0e-6; VN(Akt,i,j,k,1)=1.0e-6; VN(Akt,i,j,k,2)=0.0; break;
        case CASE_SEAMOUNT:                                                    /* Exec/Seamount/prob.cpp:167-190 */
            V(Akv,i,j,k)=1.0e-5; VN(Akt,i,j,k,0)=1.0e-6; VN(Akt,i,j,k,1)=1.0e-6; VN(Akt,i,j,k,2)=0.0; break;
        case CASE_DOUBLEGYRE:                                                  /* Exec/DoubleGyre/prob.cpp:116-138 */
            V(Akv,i,j,k)=1.0; VN(Akt,i,j,k,0)=1.0; VN(Akt,i,j,k,1)=1.0; VN(Akt,i,j,k,2)=0.0; break;
        }
    }
}

/* REMORA::set_analytical_vmix, Source/REMORA.cpp:565-573 */
static void set_analytical_vmix(OState *s)
{
    init_vmix(s);
    o_fillpatch(&s->geom,&s->Akv,F(mskr),BCV_ZVEL,0,1);
    for(int n=0;n<NCONS;n++) o_fillpatch(&s->geom,&s->Akt,F(mskr),BCV_ZVEL,0,1);
}

/* REMORA::init_gls_vmix, Initialization/REMORA_init.cpp:156-190 */
static void init_gls_vmix(OState *s)
{
    const GlsParams *G=&s->prm.gls; int N=s->N;
    fab_setval(F(tke),G->Kmin); fab_setval(F(gls),G->Pmin); fab_setval(F(Lscale),0.0);
    fab_setval(F(Akk),G->Akk_bak); fab_setval(F(Akp),G->Akp_bak); fab_setval(F(Akv),G->Akv_bak); fab_setval(F(Akt),G->Akt_bak);
    Fab Akk=F(Akk), Akp=F(Akp), Akv=F(Akv), Akt=F(Akt);
    Box b=box_slab(fab_box(Akk),2,0);
    FORBOX2D(b,i,j){
        V(Akk,i,j,0)=0.0; V(Akk,i,j,N+1)=0.0; V(Akp,i,j,0)=0.0; V(Akp,i,j,N+1)=0.0;
        V(Akv,i,j,0)=0.0; V(Akv,i,j,N+1)=0.0; V(Akt,i,j,0)=0.0; V(Akt,i,j,N+1)=0.0;      /* Akt: component 0 only */
    }
}

static void init_hmix(OState *s)
{
    const Params *P=&s->prm; Fab vp=F(visc2_p), vr=F(visc2_r), d2=F(diff2);
    if(P->hmix_constant){                                                      /* Source/REMORA.cpp:581-586 */
        fab_setval(vp,P->visc2); fab_setval(vr,P->visc2);
        for(int n=0;n<NCONS;n++) fab_setval(fab_view(d2.p+d2.sn*n,fab_box(d2),1),P->tnu2[n]);
    } else {
        Box b=box_grow(s->bx,NGROW,NGROW,0);
        double v=0.0,d=0.0;
        switch(P->prob){ case CASE_UPWELLING: case CASE_ADVECTION: case CASE_DOUBLYPERIODIC: v=5.0; d=0.0; break;
                         case CASE_SEAMOUNT: v=0.0; d=0.0; break; case CASE_DOUBLEGYRE: v=1280.0; d=1280.0; break;
                         case CASE_CHANNEL: v=2.0; d=0.0; break; }                 /* Exec/Channel_Test/prob.cpp:178-202 */
        FORBOX(b,i,j,k){ V(vp,i,j,k)=v; V(vr,i,j,k)=v; for(int n=0;n<NCONS;n++) VN(d2,i,j,k,n)=d; }
    }
    o_fillpatch(&s->geom,&s->visc2_p,F(mskr),BCV_FOEXTRAP_PERIODIC,0,1);        /* Source/REMORA.cpp:591-595 */
    o_fillpatch(&s->geom,&s->visc2_r,F(mskr),BCV_FOEXTRAP_PERIODIC,0,1);
    for(int n=0;n<NCONS;n++) o_fillpatch(&s->geom,&s->diff2,F(mskr),BCV_FOEXTRAP_PERIODIC,n,1);
}

static void init_smflux(OState *s,double time)
{
    const Params *P=&s->prm; Fab su=F(sustr), sv=F(svstr);
    switch(P->prob){
    case CASE_UPWELLING: case CASE_DOUBLYPERIODIC: {                           /* Exec/Upwelling/prob.cpp:214-255 == DoublyPeriodic/prob.cpp:274-308 */
        double pi=3.14159265359, tdays=time/(24.0*60.0*60.0), dstart=0.0, windamp=0.0;
        if(P->periodic[1]) fab_setval(su,0.0);
        else if(P->periodic[0]){
            if((tdays-dstart)<=2.0) windamp=-0.1*sin(pi*(tdays-dstart)/4.0)/P->rho0;
            else windamp=-0.1/P->rho0;
            fab_setval(su,windamp);
        }
        if(P->periodic[1]){
            if((tdays-dstart)<=2.0) windamp=-0.1*sin(pi*(tdays-dstart)/4.0)/P->rho0;
            else windamp=-0.1/P->rho0;
            fab_setval(sv,windamp);
        } else if(P->periodic[0]) fab_setval(sv,0.0);
    } break;
    case CASE_SEAMOUNT: case CASE_ADVECTION: fab_setval(su,0.0); fab_setval(sv,0.0); break;
    case CASE_CHANNEL: {                                                       /* Exec/Channel_Test/prob.cpp:204-222: oscillating along-channel wind */
        double tdays=time/(24.0*60.0*60.0), dstart=0.0, userval=0.01, fcor=1.0e-4;
        double windamp_u=userval*sin(0.9*fcor*86400.0*(tdays-dstart))/P->rho0;
        fab_setval(su,windamp_u); fab_setval(sv,0.0);
    } break;
    case CASE_DOUBLEGYRE: {                                                    /* Exec/DoubleGyre/prob.cpp:167-206 */
        double pi=3.14159265359;
        double yextent=P->prob_hi[1]-P->prob_lo[1];
        double windamp=-0.05/P->rho0, val1=2.0*pi/yextent;
        double dx1=(P->prob_hi[1]-P->prob_lo[1])/s->ny;
        Box b=box_grow(s->sustr.valid,NGROW,NGROW,0);
        FORBOX2D(b,i,j){ double y=P->prob_lo[1]+(j+0.5)*dx1; V(su,i,j,0)=windamp*cos(val1*y); }
        fab_setval(sv,0.0);
    } break;
    }
}

/* ------------------------------------------------------------------ create / init */
int oracle_omp_threads = 1;     /* see fab.h: FORBOX / FORBOX2D */
OState* o_create(const Params *prm)
{
    oracle_omp_threads = prm->threads > 1 ? prm->threads : 1;
    OState *s=(OState*)calloc(1,sizeof(OState));
    s->prm=*prm; const Params *P=&s->prm;
    s->nx=P->ncell[0]; s->ny=P->ncell[1]; s->nz=P->ncell[2]; s->N=s->nz-1;
    s->bx=box_make(0,0,0,s->nx-1,s->ny-1,s->nz-1);
    s->geom.domain=s->bx; s->geom.per[0]=P->periodic[0]; s->geom.per[1]=P->periodic[1]; s->geom.prm=&s->prm;
    o_init_bcs(&s->geom);
    s->dt=P->fixed_dt; s->t_new=0.0; s->t_old=-1.e200; s->istep=0;
    s->dxi[0]=s->nx/(P->prob_hi[0]-P->prob_lo[0]); s->dxi[1]=s->ny/(P->prob_hi[1]-P->prob_lo[1]);
    Box b3=s->bx, b2=box_slab(s->bx,2,0);
    const int G=NGROW;
    /* Initialization/REMORA_make_new_level.cpp:259-282,381-389,426-503 (no z ghost cells: none is read on the path) */
    s->cons_old=mf_alloc(b3,0,0,0,NCONS,3,3,0); s->cons_new=mf_alloc(b3,0,0,0,NCONS,3,3,0);
    s->xvel_old=mf_alloc(b3,1,0,0,1,3,3,0); s->xvel_new=mf_alloc(b3,1,0,0,1,3,3,0);
    s->yvel_old=mf_alloc(b3,0,1,0,1,3,3,0); s->yvel_new=mf_alloc(b3,0,1,0,1,3,3,0);
    s->zvel_old=mf_alloc(b3,0,0,1,1,3,3,0); s->zvel_new=mf_alloc(b3,0,0,1,1,3,3,0);
    s->Zt_avg1=mf_alloc(b2,0,0,0,1,G+1,G+1,0); s->h=mf_alloc(b2,0,0,0,2,G+1,G+1,0);
    s->ubar=mf_alloc(b2,1,0,0,3,G,G,0); s->vbar=mf_alloc(b2,0,1,0,3,G,G,0);
    s->ru=mf_alloc(b3,1,0,0,2,G,G,0); s->rv=mf_alloc(b3,0,1,0,2,G,G,0);
    s->ru2d=mf_alloc(b2,1,0,0,2,G,G,0); s->rv2d=mf_alloc(b2,0,1,0,2,G,G,0);
    s->mskr=mf_alloc(b2,0,0,0,1,G+1,G+1,0); s->msku=mf_alloc(b2,1,0,0,1,G+1,G+1,0);
    s->mskv=mf_alloc(b2,0,1,0,1,G+1,G+1,0); s->mskp=mf_alloc(b2,1,1,0,1,G+1,G+1,0);
    fab_setval(F(mskr),1.0); fab_setval(F(msku),1.0); fab_setval(F(mskv),1.0); fab_setval(F(mskp),1.0);
    s->z_w=mf_alloc(b3,0,0,1,1,G+1,G+1,0); s->z_r=mf_alloc(b3,0,0,0,1,G+1,G+1,0); s->Hz=mf_alloc(b3,0,0,0,1,G+1,G+1,0);
    s->Huon=mf_alloc(b3,1,0,0,1,G,G,0); s->Hvom=mf_alloc(b3,0,1,0,1,G,G,0);
    s->Akv=mf_alloc(b3,0,0,1,1,G,G,0); s->Akt=mf_alloc(b3,0,0,1,NCONS,G,G,0);
    s->visc2_p=mf_alloc(b3,0,0,0,1,G,G,0); s->visc2_r=mf_alloc(b3,0,0,0,1,G,G,0); s->diff2=mf_alloc(b3,0,0,0,NCONS,G,G,0);
    s->rufrc=mf_alloc(b2,1,0,0,2,G,G,0); s->rvfrc=mf_alloc(b2,0,1,0,2,G,G,0);
    s->sustr=mf_alloc(b2,1,0,0,1,G,G,0); s->svstr=mf_alloc(b2,0,1,0,1,G,G,0);
    s->rdrag=mf_alloc(b2,0,0,0,1,G,G,0);
    s->bustr=mf_alloc(b2,1,0,0,1,G,G,0); s->bvstr=mf_alloc(b2,0,1,0,1,G,G,0);
    s->DU_avg1=mf_alloc(b2,1,0,0,1,G,G,0); s->DU_avg2=mf_alloc(b2,1,0,0,1,G,G,0);
    s->DV_avg1=mf_alloc(b2,0,1,0,1,G,G,0); s->DV_avg2=mf_alloc(b2,0,1,0,1,G,G,0);
    s->rubar=mf_alloc(b2,1,0,0,4,G,G,0); s->rvbar=mf_alloc(b2,0,1,0,4,G,G,0); s->rzeta=mf_alloc(b2,0,0,0,4,G,G,0);
    s->zeta=mf_alloc(b2,0,0,0,3,G+1,G+1,0);
    s->pm=mf_alloc(b2,0,0,0,1,G+1,G+2,0); s->pn=mf_alloc(b2,0,0,0,1,G+2,G+1,0); s->fcor=mf_alloc(b2,0,0,0,1,G+1,G+1,0);
    s->sstore=mf_alloc(b3,0,0,0,NCONS,G,G,0);
    s->rhoS=mf_alloc(b2,0,0,0,1,G,G,0); s->rhoA=mf_alloc(b2,0,0,0,1,G,G,0); s->bvf=mf_alloc(b3,0,0,1,1,G,G,0);
    s->tke=mf_alloc(b3,0,0,1,3,G,G,0); s->gls=mf_alloc(b3,0,0,1,3,G,G,0); s->Lscale=mf_alloc(b3,0,0,1,1,G,G,0);
    s->Akk=mf_alloc(b3,0,0,1,1,G,G,0); s->Akp=mf_alloc(b3,0,0,1,1,G,G,0);
    fab_setval(F(tke),P->gls.Kmin); fab_setval(F(gls),P->gls.Pmin); fab_setval(F(Lscale),0.0);         /* make_new_level.cpp:517-521 */
    fab_setval(F(Akk),P->gls.Akk_bak); fab_setval(F(Akp),P->gls.Akp_bak);

    s->weight1=(double*)calloc(2*P->ndtfast+1,sizeof(double)); s->weight2=(double*)calloc(2*P->ndtfast+1,sizeof(double));
    s->nfast=o_set_weights(P->ndtfast,s->weight1,s->weight2);                  /* make_new_level.cpp:505 */
    fab_setval(F(DU_avg1),0.0); fab_setval(F(DU_avg2),0.0); fab_setval(F(DV_avg1),0.0); fab_setval(F(DV_avg2),0.0);
    fab_setval(F(rubar),0.0); fab_setval(F(rvbar),0.0); fab_setval(F(rzeta),0.0);
    fab_setval(F(rdrag),P->rdrag);

    /* REMORA::init_only, Source/REMORA.cpp:608-702 */
    fab_setval(F(cons_new),0.0); fab_setval(F(xvel_new),0.0); fab_setval(F(yvel_new),0.0); fab_setval(F(zvel_new),0.0);
    fab_setval(F(xvel_old),0.0); fab_setval(F(yvel_old),0.0); fab_setval(F(zvel_old),0.0);
    fab_setval(F(ru),0.0); fab_setval(F(rv),0.0); fab_setval(F(ru2d),0.0); fab_setval(F(rv2d),0.0);
    init_bathymetry(s);                                                        /* set_bathymetry :464-526 */
    o_fill_boundary(&s->geom,&s->h,0,2); o_fill_boundary(&s->geom,&s->h,0,2);
    fab_setval(F(zeta),0.0);                                                   /* set_zeta :444-461 */
    o_fill_boundary(&s->geom,&s->zeta,0,3);
    { Box b=box_slab(box_grow(s->bx,G+1,G+1,0),2,0); Fab Zt=F(Zt_avg1), ze=F(zeta);
      FORBOX2D(b,i,j) V(Zt,i,j,0)=VN(ze,i,j,0,0); }                            /* REMORA_init.cpp:76-95 */
    stretch_transform(s);
    init_prob(s);                                                              /* init_custom, REMORA_init.cpp:11-46 */
    fab_setval(F(pm),s->dxi[0]); fab_setval(F(pn),s->dxi[1]);                  /* set_pm_pn, make_new_level.cpp:548-553 */
    { /* set_2darrays, REMORA_init.cpp:97-154 */
      fab_setval(F(ubar),0.0); fab_setval(F(vbar),0.0);
      Fab ub=F(ubar), vb=F(vbar), Hz=F(Hz), u=F(xvel_new), v=F(yvel_new);
      Box ub2=box_slab(box_grow(s->xvel_new.valid,G,G,0),2,0), vb2=box_slab(box_grow(s->yvel_new.valid,G,G,0),2,0);
      FORBOX2D(ub2,i,j){ double CF=0.,sum=0.;
          for(int k=0;k<=s->N;k++){ double a=0.5*(V(Hz,i,j,k)+V(Hz,i-1,j,k)); sum+=a; CF+=a*V(u,i,j,k); }
          VN(ub,i,j,0,0)=CF/sum; }
      FORBOX2D(vb2,i,j){ double CF=0.,sum=0.;
          for(int k=0;k<=s->N;k++){ double a=0.5*(V(Hz,i,j,k)+V(Hz,i,j-1,k)); sum+=a; CF+=a*V(v,i,j,k); }
          VN(vb,i,j,0,0)=CF/sum; }
      o_fillpatch(&s->geom,&s->ubar,F(msku),BCV_UBAR,0,1);
      o_fillpatch(&s->geom,&s->vbar,F(mskv),BCV_VBAR,0,1);
    }
    if(P->gls.on) init_gls_vmix(s); else set_analytical_vmix(s);               /* init_set_vmix, REMORA.cpp:550-562 */
    init_hmix(s);                                                              /* set_hmixcoef */
    if(P->use_coriolis){                                                       /* set_coriolis :528-549, REMORA_init.cpp:48-74 */
        Fab fc=F(fcor); Box b=fab_box(fc);
        double Esize=P->prob_hi[1]-P->prob_lo[1], dx1=(P->prob_hi[1]-P->prob_lo[1])/s->ny;
        FORBOX2D(b,i,j){ double y=P->prob_lo[1]+(j+0.5)*dx1; V(fc,i,j,0)=P->coriolis_f0+P->coriolis_beta*(y-0.5*Esize); }
        o_fill_boundary(&s->geom,&s->fcor,0,1);
        o_fillpatch(&s->geom,&s->fcor,F(mskr),BCV_FOEXTRAP,0,1);
        o_fill_boundary(&s->geom,&s->fcor,0,1);
    }
    init_smflux(s,0.0);
    /* REMORA::InitData, Source/REMORA.cpp:293-304 */
    o_fillpatch_calc(&s->geom,&s->cons_new,F(mskr),BCV_CONS,0,NCONS,F(cons_new),F(msku),F(mskv));   /* mf_calc = the array itself */
    o_fillpatch_calc(&s->geom,&s->xvel_new,F(msku),BCV_XVEL,0,1,F(xvel_new),F(msku),F(mskv));
    o_fillpatch_calc(&s->geom,&s->yvel_new,F(mskv),BCV_YVEL,0,1,F(yvel_new),F(msku),F(mskv));
    o_fillpatch(&s->geom,&s->zvel_new,F(mskr),BCV_ZVEL,0,1);
    memcpy(F(cons_old).p,F(cons_new).p,sizeof(double)*fab_size(F(cons_new)));
    memcpy(F(xvel_old).p,F(xvel_new).p,sizeof(double)*fab_size(F(xvel_new)));
    memcpy(F(yvel_old).p,F(yvel_new).p,sizeof(double)*fab_size(F(yvel_new)));
    memcpy(F(zvel_old).p,F(zvel_new).p,sizeof(double)*fab_size(F(zvel_new)));
    return s;
}

void o_destroy(OState *s)
{
    MF *all[]={&s->cons_old,&s->cons_new,&s->xvel_old,&s->xvel_new,&s->yvel_old,&s->yvel_new,&s->zvel_old,&s->zvel_new,
        &s->Zt_avg1,&s->h,&s->ubar,&s->vbar,&s->ru,&s->rv,&s->ru2d,&s->rv2d,&s->mskr,&s->msku,&s->mskv,&s->mskp,
        &s->z_w,&s->z_r,&s->Hz,&s->Huon,&s->Hvom,&s->Akv,&s->Akt,&s->visc2_p,&s->visc2_r,&s->diff2,
        &s->rufrc,&s->rvfrc,&s->sustr,&s->svstr,&s->rdrag,&s->bustr,&s->bvstr,&s->DU_avg1,&s->DU_avg2,&s->DV_avg1,&s->DV_avg2,
        &s->rubar,&s->rvbar,&s->rzeta,&s->zeta,&s->pm,&s->pn,&s->fcor,&s->sstore,&s->rhoS,&s->rhoA,&s->bvf,
        &s->tke,&s->gls,&s->Lscale,&s->Akk,&s->Akp};
    for(unsigned q=0;q<sizeof(all)/sizeof(all[0]);q++) mf_free(all[q]);
    free(s->weight1); free(s->weight2); free(s);
}

/* ------------------------------------------------------------------ prestep
 * TimeIntegration/REMORA_prestep.cpp:36-185 */
static void prestep(OState *s,MF *mf_W,MF *mf_DC,int iic,int ntfirst,int nnew,int nstp,int nrhs,int N,double dt_lev)
{
    const Geom *g=&s->geom;
    Box bx=s->bx, gbx=box_grow(bx,3,3,3);
    Box tbxp1=box_grow(bx,NGROW-1,NGROW-1,0), tbxp2=box_grow(bx,NGROW,NGROW,0);
    Fab cache=fab_alloc(tbxp2,NCONS);
    fab_copy(cache,0,F(cons_new),0,NCONS,tbxp2);                               /* :66-67 */
    Fab FC=fab_alloc(box_nodal(tbxp2,2),1);
    Fab stflux=fab_alloc(tbxp2,1), btflux=fab_alloc(tbxp2,1);
    fab_setval(stflux,0.0); fab_setval(btflux,0.0);
    double lambda=1.0;
    for(int n=0;n<NCONS;n++)                                                   /* :157-163 */
        o_prestep_t_advection(g,bx,gbx,fab_comp(F(cons_old),n),fab_comp(cache,n),F(Hz),F(Huon),F(Hvom),
                              mf_W->f,mf_DC->f,FC,fab_comp(F(sstore),n),F(z_w),F(h),F(pm),F(pn),F(msku),F(mskv),
                              iic,ntfirst,nrhs,N,dt_lev);
    for(int n=0;n<NCONS;n++)                                                   /* :166-170 (Akt passed whole: component 0) */
        o_prestep_diffusion(g,bx,gbx,0,0,fab_comp(F(cons_new),n),fab_comp(F(cons_old),n),F(ru),F(Hz),F(Akt),mf_DC->f,FC,
                            stflux,btflux,F(z_r),F(pm),F(pn),iic,iic,nnew,nstp,nrhs,N,lambda,dt_lev);
    o_prestep_diffusion(g,tbxp1,gbx,1,0,F(xvel_new),F(xvel_old),F(ru),F(Hz),F(Akv),mf_DC->f,FC,F(sustr),F(bustr),
                        F(z_r),F(pm),F(pn),iic,ntfirst,nnew,nstp,nrhs,N,lambda,dt_lev);                /* :179-180 */
    o_prestep_diffusion(g,tbxp1,gbx,0,1,F(yvel_new),F(yvel_old),F(rv),F(Hz),F(Akv),mf_DC->f,FC,F(svstr),F(bvstr),
                        F(z_r),F(pm),F(pn),iic,ntfirst,nnew,nstp,nrhs,N,lambda,dt_lev);                /* :182-183 */
    fab_free(&cache); fab_free(&FC); fab_free(&stflux); fab_free(&btflux);
}

/* ------------------------------------------------------------------ setup_step
 * TimeIntegration/REMORA_setup_step.cpp:7-369 */
void o_setup_step(OState *s)
{
    const Geom *g=&s->geom; const Params *P=&s->prm;
    double dt_lev=s->dt; int N=s->N;
    Box bx=s->bx;
    o_fill_boundary(g,&s->cons_old,0,NCONS);                                   /* :25-28 */
    o_fill_boundary(g,&s->xvel_old,0,1);
    o_fill_boundary(g,&s->yvel_old,0,1);
    o_fillpatch(g,&s->zvel_old,F(mskr),BCV_ZVEL,0,1);
    const int nrhs=0; int nstp=0;
    MF mf_DC=mf_alloc(bx,0,0,0,1,NGROW,NGROW,NGROW-1);
    MF mf_rho=mf_alloc(bx,0,0,0,1,NGROW,NGROW,0);
    fab_setval_box(mf_rho.f,box_grow(bx,NGROW-1,NGROW-1,0),0.0);               /* :89-91 (0.e34 == 0) */
    fab_setval_box(F(rhoS),box_grow(box_slab(bx,2,0),NGROW-1,NGROW-1,0),0.0);
    fab_setval_box(F(rhoA),box_grow(box_slab(bx,2,0),NGROW-1,NGROW-1,0),0.0);
    fab_setval(mf_DC.f,0.0);
    o_fill_boundary(g,&s->cons_new,0,NCONS);                                   /* :95-97 */
    o_fill_boundary(g,&s->xvel_new,0,1);
    o_fill_boundary(g,&s->yvel_new,0,1);
    fab_setval(F(rufrc),0.0); fab_setval(F(rvfrc),0.0);
    int iic=s->istep, ntfirst=0;
    if(iic==ntfirst){                                                          /* :105-110 */
        memcpy(F(cons_new).p,F(cons_old).p,sizeof(double)*fab_size(F(cons_new)));
        memcpy(F(xvel_new).p,F(xvel_old).p,sizeof(double)*fab_size(F(xvel_new)));
        memcpy(F(yvel_new).p,F(yvel_old).p,sizeof(double)*fab_size(F(yvel_new)));
        memcpy(F(zvel_new).p,F(zvel_old).p,sizeof(double)*fab_size(F(zvel_new)));
    }
    init_smflux(s,s->t_old);                                                   /* :111 */
    {                                                                          /* :115-140 */
        Fab rdrag=F(rdrag), bustr=F(bustr), bvstr=F(bvstr), uold=F(xvel_old), vold=F(yvel_old);
        Box ub=box_slab(box_grow(box_nodal(bx,0),NGROW-1,NGROW-1,0),2,0), vb=box_slab(box_grow(box_nodal(bx,1),NGROW-1,NGROW-1,0),2,0);
        FORBOX2D(ub,i,j) V(bustr,i,j,0)=0.5*(V(rdrag,i-1,j,0)+V(rdrag,i,j,0))*(V(uold,i,j,0));
        FORBOX2D(vb,i,j) V(bvstr,i,j,0)=0.5*(V(rdrag,i,j-1,0)+V(rdrag,i,j,0))*(V(vold,i,j,0));
        o_fillpatch(g,&s->bustr,F(msku),BCV_U2D_SIMPLE,0,1);
        o_fillpatch(g,&s->bvstr,F(mskv),BCV_V2D_SIMPLE,0,1);
    }
    {                                                                          /* :142-201 */
        Fab Hz=F(Hz), Huon=F(Huon), Hvom=F(Hvom), uold=F(xvel_old), vold=F(yvel_old), pm=F(pm), pn=F(pn);
        Box bu=fab_box(Huon), bv=fab_box(Hvom);
        FORBOX(bu,i,j,k){ double on_u=2.0/(V(pn,i-1,j,0)+V(pn,i,j,0)); V(Huon,i,j,k)=0.5*(V(Hz,i,j,k)+V(Hz,i-1,j,k))*V(uold,i,j,k)*on_u; }
        FORBOX(bv,i,j,k){ double om_v=2.0/(V(pm,i,j-1,0)+V(pm,i,j,0)); V(Hvom,i,j,k)=0.5*(V(Hz,i,j,k)+V(Hz,i,j-1,k))*V(vold,i,j,k)*om_v; }
        o_rho_eos(g,box_grow(bx,NGROW,NGROW,0),F(cons_old),mf_rho.f,F(rhoA),F(rhoS),F(bvf),Hz,F(z_w),F(z_r),F(h),F(mskr),N);
    }
    if(!P->gls.on) set_analytical_vmix(s);                                     /* :202-205 */
    { Box b=box_slab(box_grow(bx,NGROW+1,NGROW+1,0),2,0); Fab ze=F(zeta), Zt=F(Zt_avg1);   /* set_zeta_to_Ztavg :208 */
      for(int n=0;n<3;n++) FORBOX2D(b,i,j) VN(ze,i,j,0,n)=V(Zt,i,j,0); }
    MF mf_W=mf_alloc(bx,0,0,1,1,NGROW+1,NGROW+1,0); fab_setval(mf_W.f,0.0);    /* :210-211 */
    if(P->use_prestep) prestep(s,&mf_W,&mf_DC,iic,ntfirst,0,nstp,nrhs,N,dt_lev);
    o_fill_boundary(g,&mf_W,0,1);                                              /* :226-227 */
    o_physbc(g,&mf_W,F(mskr),0,1,BCV_ZVEL,0);
    {                                                                          /* :229-350 */
        Box tbxp1=box_grow(bx,NGROW-1,NGROW-1,0), tbxp2=box_grow(bx,NGROW,NGROW,0);
        Box xbx=box_nodal(bx,0), ybx=box_nodal(bx,1), gbx1=tbxp1, gbx2=tbxp2;
        Fab FC=fab_alloc(box_nodal(tbxp2,2),1);
        Fab fomn=fab_alloc(box_slab(tbxp2,2,0),1);
        if(P->use_coriolis){ Box b=box_slab(tbxp2,2,0); Fab fc=F(fcor), pm=F(pm), pn=F(pn);
            FORBOX2D(b,i,j) V(fomn,i,j,0)=V(fc,i,j,0)*(1.0/(V(pm,i,j,0)*V(pn,i,j,0))); }
        fab_setval_box(FC,gbx2,0.0);
        o_prsgrd(g,tbxp1,gbx1,xbx,ybx,F(ru),F(rv),F(pn),F(pm),mf_rho.f,FC,F(Hz),F(z_r),F(z_w),F(msku),F(mskv),nrhs,N);
        int ncomp=P->use_salt?2:1;
        o_t3dmix(g,bx,F(cons_new),F(cons_old),F(diff2),F(Hz),F(pm),F(pn),F(msku),F(mskv),dt_lev,ncomp);
        o_t3dmix(g,bx,fab_comp(F(cons_new),2),fab_comp(F(cons_old),2),fab_comp(F(diff2),2),F(Hz),F(pm),F(pn),F(msku),F(mskv),dt_lev,1);
        if(P->use_coriolis) o_coriolis(xbx,ybx,F(xvel_old),F(yvel_old),F(ru),F(rv),F(Hz),fomn,nrhs,nrhs);
        o_rhs_uv_3d(g,xbx,ybx,F(xvel_old),F(yvel_old),F(ru),F(rv),F(rufrc),F(rvfrc),F(sustr),F(svstr),F(bustr),F(bvstr),
                    F(Huon),F(Hvom),F(pm),F(pn),mf_W.f,FC,nrhs,N);
        if(P->use_uv3dmix)
            o_uv3dmix(xbx,ybx,F(xvel_new),F(yvel_new),F(xvel_old),F(yvel_old),F(rufrc),F(rvfrc),F(visc2_p),F(visc2_r),F(Hz),
                      F(pm),F(pn),F(mskp),nrhs,0,dt_lev);
        fab_free(&FC); fab_free(&fomn);
    }
    if(P->gls.on){                                                             /* :352-357 */
        int nnew_g=(iic+1)%2, nstp_g=iic%2;
        o_gls_prestep(g,bx,F(gls),F(tke),mf_W.f,F(Huon),F(Hvom),F(Hz),F(pm),F(pn),F(msku),F(mskv),nstp_g,nnew_g,iic,ntfirst,N,dt_lev);
        for(int ic=0;ic<3;ic++){                                               /* REMORA_gls.cpp:212-215 */
            o_fillpatch(g,&s->tke,F(mskr),BCV_ZVEL,ic,1);
            o_fillpatch(g,&s->gls,F(mskr),BCV_ZVEL,ic,1);
        }
    }
    o_fillpatch_calc(g,&s->sstore,F(mskr),BCV_CONS,0,NCONS,F(cons_old),F(msku),F(mskv));   /* :363, mf_calc = cons_old */
    o_fill_boundary(g,&s->Huon,0,1);                                           /* :367-368 */
    o_fill_boundary(g,&s->Hvom,0,1);
    mf_free(&mf_W); mf_free(&mf_DC); mf_free(&mf_rho);
}

/* ------------------------------------------------------------------ advance_2d
 * TimeIntegration/REMORA_advance_2d.cpp:41-757 */
static void advance_2d(OState *s,double dtfast_lev,int predictor_2d_step,int first_2d_step,int my_iif,int *next_indx1)
{
    const Geom *g=&s->geom; const Params *P=&s->prm;
    int iic=s->istep; const int nnew=0, nstp=0; int ntfirst=0;
    int knew=3;
    int krhs=(my_iif+iic)%2+1;
    int kstp= my_iif<=1 ? iic%2+1 : (iic%2+my_iif%2+1)%2+1;
    int indx1=krhs;
    if(predictor_2d_step) *next_indx1=3-indx1;
    else { knew=*next_indx1; kstp=3-knew; krhs=3; }
    int ptsk=3-kstp;
    knew-=1; krhs-=1; kstp-=1; ptsk-=1;
    int nfast=s->nfast;
    const double *weight1=s->weight1,*weight2=s->weight2;

    Box bx=s->bx;                                                              /* 3-D tile box of the (3-D allocated) rhoS */
    Box b2=box_slab(bx,2,0);
    Box tbxp1=box_grow(b2,NGROW-1,NGROW-1,0), tbxp2=box_grow(b2,NGROW,NGROW,0), tbxp3=box_grow(b2,NGROW+1,NGROW+1,0);
    Box gbx1=tbxp1, gbx3=tbxp3;
    Box xgbx2=box_grow(box_nodal(b2,0),NGROW,NGROW,0), ygbx2=box_grow(box_nodal(b2,1),NGROW,NGROW,0);
    Box xbxD=box_nodal(b2,0), ybxD=box_nodal(b2,1);

    MF mf_DUon=mf_alloc(b2,1,0,0,1,NGROW,NGROW,0), mf_DVom=mf_alloc(b2,0,1,0,1,NGROW,NGROW,0);
    Fab DUon=mf_DUon.f, DVom=mf_DVom.f;
    Fab ubar=F(ubar), vbar=F(vbar), zeta=F(zeta), h=F(h), pm=F(pm), pn=F(pn);
    Fab Drhs=fab_alloc(tbxp3,1);
    FORBOX2D(tbxp3,i,j) V(Drhs,i,j,0)=VN(zeta,i,j,0,krhs)+V(h,i,j,0);          /* :149-152 */
    FORBOX2D(xgbx2,i,j){                                                       /* :154-159 */
        double on_u=2.0/(V(pn,i,j,0)+V(pn,i-1,j,0));
        double cff1=0.5*on_u*(V(Drhs,i,j,0)+V(Drhs,i-1,j,0));
        V(DUon,i,j,0)=VN(ubar,i,j,0,krhs)*cff1;
    }
    FORBOX2D(ygbx2,i,j){                                                       /* :161-166 */
        double om_v=2.0/(V(pm,i,j,0)+V(pm,i,j-1,0));
        double cff1=0.5*om_v*(V(Drhs,i,j,0)+V(Drhs,i,j-1,0));
        V(DVom,i,j,0)=VN(vbar,i,j,0,krhs)*cff1;
    }
    o_fill_boundary(g,&mf_DUon,0,1);                                           /* :170-171 */
    o_fill_boundary(g,&mf_DVom,0,1);

    Fab rhoS=F(rhoS), rhoA=F(rhoA), rufrc=F(rufrc), rvfrc=F(rvfrc), Zt_avg1=F(Zt_avg1);
    Fab DU_avg1=F(DU_avg1), DU_avg2=F(DU_avg2), DV_avg1=F(DV_avg1), DV_avg2=F(DV_avg2);
    Fab ru2d=F(ru2d), rv2d=F(rv2d), rubar=F(rubar), rvbar=F(rvbar), rzeta=F(rzeta);
    Fab fcor=F(fcor), mskr=F(mskr), msku=F(msku), mskv=F(mskv);

    Fab fomn=fab_alloc(tbxp2,1), Dnew=fab_alloc(tbxp2,1), zwrk=fab_alloc(tbxp1,1), gzeta=fab_alloc(tbxp1,1),
        gzeta2=fab_alloc(tbxp1,1), gzetaSA=fab_alloc(tbxp1,1), Dstp=fab_alloc(tbxp3,1),
        rhs_ubar=fab_alloc(xbxD,1), rhs_vbar=fab_alloc(ybxD,1), rhs_zeta=fab_alloc(tbxp1,1), zeta_new=fab_alloc(tbxp1,1);
    fab_setval(rhs_ubar,0.0); fab_setval(rhs_vbar,0.0);                        /* :271-279 */
    if(P->use_coriolis) FORBOX2D(tbxp2,i,j) V(fomn,i,j,0)=V(fcor,i,j,0)*(1.0/(V(pm,i,j,0)*V(pn,i,j,0)));
    FORBOX2D(tbxp3,i,j) V(Drhs,i,j,0)=VN(zeta,i,j,0,krhs)+V(h,i,j,0);          /* :288-291 */

    if(predictor_2d_step){                                                     /* :293-356 */
        if(first_2d_step){
            double cff2=(-1.0/12.0)*weight2[my_iif+1];
            FORBOX2D(gbx3,i,j) V(Zt_avg1,i,j,0)=0.0;
            FORBOX2D(xgbx2,i,j){ V(DU_avg1,i,j,0)=0.0; V(DU_avg2,i,j,0)=cff2*V(DUon,i,j,0); }
            FORBOX2D(ygbx2,i,j){ V(DV_avg1,i,j,0)=0.0; V(DV_avg2,i,j,0)=cff2*V(DVom,i,j,0); }
        } else {
            double cff1_wt1=weight1[my_iif-1];
            double cff2_wt1=(8.0/12.0)*weight2[my_iif]-(1.0/12.0)*weight2[my_iif+1];
            FORBOX2D(gbx3,i,j) V(Zt_avg1,i,j,0)+=cff1_wt1*VN(zeta,i,j,0,krhs);
            FORBOX2D(xgbx2,i,j){ V(DU_avg1,i,j,0)+=cff1_wt1*V(DUon,i,j,0); V(DU_avg2,i,j,0)+=cff2_wt1*V(DUon,i,j,0); }
            FORBOX2D(ygbx2,i,j){ V(DV_avg1,i,j,0)+=cff1_wt1*V(DVom,i,j,0); V(DV_avg2,i,j,0)+=cff2_wt1*V(DVom,i,j,0); }
        }
    } else {
        double cff2_wt2;
        if(first_2d_step) cff2_wt2=weight2[my_iif]; else cff2_wt2=5.0/12.0*weight2[my_iif];
        FORBOX2D(xgbx2,i,j) V(DU_avg2,i,j,0)=V(DU_avg2,i,j,0)+cff2_wt2*V(DUon,i,j,0);
        FORBOX2D(ygbx2,i,j) V(DV_avg2,i,j,0)=V(DV_avg2,i,j,0)+cff2_wt2*V(DVom,i,j,0);
    }

    if(my_iif<nfast){                                                          /* :362-363 */
        double fac=1000.0/1025.0;                                              /* :379 */
        if(my_iif==0){                                                         /* :381-396 */
            double cff1=dtfast_lev;
            FORBOX2D(tbxp1,i,j){
                V(rhs_zeta,i,j,0)=(V(DUon,i,j,0)-V(DUon,i+1,j,0))+(V(DVom,i,j,0)-V(DVom,i,j+1,0));
                V(zeta_new,i,j,0)=(VN(zeta,i,j,0,kstp)+V(pm,i,j,0)*V(pn,i,j,0)*cff1*V(rhs_zeta,i,j,0))*V(mskr,i,j,0);
                V(Dnew,i,j,0)=V(zeta_new,i,j,0)+V(h,i,j,0);
                V(zwrk,i,j,0)=0.5*(VN(zeta,i,j,0,kstp)+V(zeta_new,i,j,0));
                V(gzeta,i,j,0)=(fac+V(rhoS,i,j,0))*V(zwrk,i,j,0);
                V(gzeta2,i,j,0)=V(gzeta,i,j,0)*V(zwrk,i,j,0);
                V(gzetaSA,i,j,0)=V(zwrk,i,j,0)*(V(rhoS,i,j,0)-V(rhoA,i,j,0));
            }
        } else if(predictor_2d_step){                                          /* :398-417 */
            double cff1=2.0*dtfast_lev, cff4=4.0/25.0, cff5=1.0-2.0*cff4;
            FORBOX2D(tbxp1,i,j){
                V(rhs_zeta,i,j,0)=(V(DUon,i,j,0)-V(DUon,i+1,j,0))+(V(DVom,i,j,0)-V(DVom,i,j+1,0));
                V(zeta_new,i,j,0)=(VN(zeta,i,j,0,kstp)+V(pm,i,j,0)*V(pn,i,j,0)*cff1*V(rhs_zeta,i,j,0))*V(mskr,i,j,0);
                V(Dnew,i,j,0)=V(zeta_new,i,j,0)+V(h,i,j,0);
                V(zwrk,i,j,0)=cff5*VN(zeta,i,j,0,krhs)+cff4*(VN(zeta,i,j,0,kstp)+V(zeta_new,i,j,0));
                V(gzeta,i,j,0)=(fac+V(rhoS,i,j,0))*V(zwrk,i,j,0);
                V(gzeta2,i,j,0)=V(gzeta,i,j,0)*V(zwrk,i,j,0);
                V(gzetaSA,i,j,0)=V(zwrk,i,j,0)*(V(rhoS,i,j,0)-V(rhoA,i,j,0));
            }
        } else {                                                               /* :419-443 */
            double cff1=dtfast_lev*5.0/12.0, cff2=dtfast_lev*8.0/12.0, cff3=dtfast_lev*1.0/12.0, cff4=2.0/5.0, cff5=1.0-cff4;
            FORBOX2D(tbxp1,i,j){
                double cff=cff1*((V(DUon,i,j,0)-V(DUon,i+1,j,0))+(V(DVom,i,j,0)-V(DVom,i,j+1,0)));
                V(zeta_new,i,j,0)=VN(zeta,i,j,0,kstp)+
                    V(pm,i,j,0)*V(pn,i,j,0)*(cff+cff2*VN(rzeta,i,j,0,kstp)-cff3*VN(rzeta,i,j,0,ptsk));
                V(zeta_new,i,j,0)*=V(mskr,i,j,0);
                V(Dnew,i,j,0)=V(zeta_new,i,j,0)+V(h,i,j,0);
                V(zwrk,i,j,0)=cff5*V(zeta_new,i,j,0)+cff4*VN(zeta,i,j,0,krhs);
                V(gzeta,i,j,0)=(fac+V(rhoS,i,j,0))*V(zwrk,i,j,0);
                V(gzeta2,i,j,0)=V(gzeta,i,j,0)*V(zwrk,i,j,0);
                V(gzetaSA,i,j,0)=V(zwrk,i,j,0)*(V(rhoS,i,j,0)-V(rhoA,i,j,0));
            }
        }
        FORBOX2D(gbx1,i,j) VN(zeta,i,j,0,knew)=V(zeta_new,i,j,0);              /* :450-454 */
        if(predictor_2d_step) FORBOX2D(gbx1,i,j) VN(rzeta,i,j,0,krhs)=V(rhs_zeta,i,j,0);   /* :459-464 */

        double cff1=0.5*P->g, cff2=1.0/3.0;                                    /* :479-507 */
        FORBOX2D(xbxD,i,j){
            double on_u=2.0/(V(pn,i,j,0)+V(pn,i-1,j,0));
            V(rhs_ubar,i,j,0)=cff1*on_u*
                ((V(h,i-1,j,0)+V(h,i,j,0))*
                 (V(gzeta,i-1,j,0)-V(gzeta,i,j,0))+
                 (V(h,i-1,j,0)-V(h,i,j,0))*
                 (V(gzetaSA,i-1,j,0)+V(gzetaSA,i,j,0)+
                  cff2*(V(rhoA,i-1,j,0)-V(rhoA,i,j,0))*
                       (V(zwrk,i-1,j,0)-V(zwrk,i,j,0)))+
                 (V(gzeta2,i-1,j,0)-V(gzeta2,i,j,0)));
        }
        FORBOX2D(ybxD,i,j){
            double om_v=2.0/(V(pm,i,j,0)+V(pm,i,j-1,0));
            V(rhs_vbar,i,j,0)=cff1*om_v*
                ((V(h,i,j-1,0)+V(h,i,j,0))*
                 (V(gzeta,i,j-1,0)-V(gzeta,i,j,0))+
                 (V(h,i,j-1,0)-V(h,i,j,0))*
                 (V(gzetaSA,i,j-1,0)+V(gzetaSA,i,j,0)+
                  cff2*(V(rhoA,i,j-1,0)-V(rhoA,i,j,0))*
                       (V(zwrk,i,j-1,0)-V(zwrk,i,j,0)))+
                 (V(gzeta2,i,j-1,0)-V(gzeta2,i,j,0)));
        }
        o_rhs_uv_2d(g,xbxD,ybxD,ubar,vbar,rhs_ubar,rhs_vbar,DUon,DVom,krhs);   /* :518 */
        if(P->use_coriolis) o_coriolis(xbxD,ybxD,ubar,vbar,rhs_ubar,rhs_vbar,Drhs,fomn,krhs,0);       /* :530 */
        o_uv3dmix(xbxD,ybxD,ubar,vbar,ubar,vbar,rhs_ubar,rhs_vbar,F(visc2_p),F(visc2_r),Drhs,pm,pn,F(mskp),krhs,nnew,0.0);  /* :536-538 */

        if(first_2d_step&&predictor_2d_step){                                  /* :543-624 */
            if(iic==ntfirst){
                FORBOX2D(xbxD,i,j){ V(rufrc,i,j,0)-=V(rhs_ubar,i,j,0); V(rhs_ubar,i,j,0)+=V(rufrc,i,j,0); VN(ru2d,i,j,0,nstp)=V(rufrc,i,j,0); }
                FORBOX2D(ybxD,i,j){ V(rvfrc,i,j,0)-=V(rhs_vbar,i,j,0); V(rhs_vbar,i,j,0)+=V(rvfrc,i,j,0); VN(rv2d,i,j,0,nstp)=V(rvfrc,i,j,0); }
            } else if(iic==(ntfirst+1)){
                FORBOX2D(xbxD,i,j){
                    V(rufrc,i,j,0)=V(rufrc,i,j,0)-V(rhs_ubar,i,j,0);
                    V(rhs_ubar,i,j,0)=V(rhs_ubar,i,j,0)+1.5*V(rufrc,i,j,0)-0.5*VN(ru2d,i,j,0,0);
                    VN(ru2d,i,j,0,1)=V(rufrc,i,j,0);
                    double r_swap=VN(ru2d,i,j,0,1); VN(ru2d,i,j,0,1)=VN(ru2d,i,j,0,0); VN(ru2d,i,j,0,0)=r_swap;
                }
                FORBOX2D(ybxD,i,j){
                    V(rvfrc,i,j,0)=V(rvfrc,i,j,0)-V(rhs_vbar,i,j,0);
                    V(rhs_vbar,i,j,0)=V(rhs_vbar,i,j,0)+1.5*V(rvfrc,i,j,0)-0.5*VN(rv2d,i,j,0,0);
                    VN(rv2d,i,j,0,1)=V(rvfrc,i,j,0);
                    double r_swap=VN(rv2d,i,j,0,1); VN(rv2d,i,j,0,1)=VN(rv2d,i,j,0,0); VN(rv2d,i,j,0,0)=r_swap;
                }
            } else {
                double c1=23.0/12.0, c2=16.0/12.0, c3=5.0/12.0;
                FORBOX2D(xbxD,i,j){
                    V(rufrc,i,j,0)=V(rufrc,i,j,0)-V(rhs_ubar,i,j,0);
                    V(rhs_ubar,i,j,0)=V(rhs_ubar,i,j,0)+c1*V(rufrc,i,j,0)-c2*VN(ru2d,i,j,0,0)+c3*VN(ru2d,i,j,0,1);
                    VN(ru2d,i,j,0,1)=V(rufrc,i,j,0);
                    double r_swap=VN(ru2d,i,j,0,1); VN(ru2d,i,j,0,1)=VN(ru2d,i,j,0,0); VN(ru2d,i,j,0,0)=r_swap;
                }
                FORBOX2D(ybxD,i,j){
                    V(rvfrc,i,j,0)=V(rvfrc,i,j,0)-V(rhs_vbar,i,j,0);
                    V(rhs_vbar,i,j,0)=V(rhs_vbar,i,j,0)+c1*V(rvfrc,i,j,0)-c2*VN(rv2d,i,j,0,0)+c3*VN(rv2d,i,j,0,1);
                    VN(rv2d,i,j,0,1)=V(rvfrc,i,j,0);
                    double r_swap=VN(rv2d,i,j,0,1); VN(rv2d,i,j,0,1)=VN(rv2d,i,j,0,0); VN(rv2d,i,j,0,0)=r_swap;
                }
            }
        } else {
            FORBOX2D(xbxD,i,j) V(rhs_ubar,i,j,0)+=V(rufrc,i,j,0);
            FORBOX2D(ybxD,i,j) V(rhs_vbar,i,j,0)+=V(rvfrc,i,j,0);
        }

        FORBOX2D(tbxp3,i,j) V(Dstp,i,j,0)=VN(zeta,i,j,0,kstp)+V(h,i,j,0);      /* :633-636 */
        if(my_iif==0 || predictor_2d_step){                                    /* :643-684 (two identical bodies, different cff1) */
            double c1=(my_iif==0)?0.5*dtfast_lev:dtfast_lev;
            FORBOX2D(xbxD,i,j){
                double cff=(V(pm,i,j,0)+V(pm,i-1,j,0))*(V(pn,i,j,0)+V(pn,i-1,j,0));
                double Dnew_avg=1.0/(V(Dnew,i,j,0)+V(Dnew,i-1,j,0));
                VN(ubar,i,j,0,knew)=(VN(ubar,i,j,0,kstp)*(V(Dstp,i,j,0)+V(Dstp,i-1,j,0))+
                                     cff*c1*V(rhs_ubar,i,j,0))*Dnew_avg*V(msku,i,j,0);
            }
            FORBOX2D(ybxD,i,j){
                double cff=(V(pm,i,j,0)+V(pm,i,j-1,0))*(V(pn,i,j,0)+V(pn,i,j-1,0));
                double Dnew_avg=1.0/(V(Dnew,i,j,0)+V(Dnew,i,j-1,0));
                VN(vbar,i,j,0,knew)=(VN(vbar,i,j,0,kstp)*(V(Dstp,i,j,0)+V(Dstp,i,j-1,0))+
                                     cff*c1*V(rhs_vbar,i,j,0))*Dnew_avg*V(mskv,i,j,0);
            }
        } else {                                                               /* :686-713 */
            double c1=0.5*dtfast_lev*5.0/12.0, c2=0.5*dtfast_lev*8.0/12.0, c3=0.5*dtfast_lev*1.0/12.0;
            FORBOX2D(xbxD,i,j){
                double cff=(V(pm,i,j,0)+V(pm,i-1,j,0))*(V(pn,i,j,0)+V(pn,i-1,j,0));
                double Dnew_avg=1.0/(V(Dnew,i,j,0)+V(Dnew,i-1,j,0));
                VN(ubar,i,j,0,knew)=(VN(ubar,i,j,0,kstp)*(V(Dstp,i,j,0)+V(Dstp,i-1,j,0))+
                                     cff*(c1*V(rhs_ubar,i,j,0)+c2*VN(rubar,i,j,0,kstp)-c3*VN(rubar,i,j,0,ptsk)))*Dnew_avg*V(msku,i,j,0);
            }
            FORBOX2D(ybxD,i,j){
                double cff=(V(pm,i,j,0)+V(pm,i,j-1,0))*(V(pn,i,j,0)+V(pn,i,j-1,0));
                double Dnew_avg=1.0/(V(Dnew,i,j,0)+V(Dnew,i,j-1,0));
                VN(vbar,i,j,0,knew)=(VN(vbar,i,j,0,kstp)*(V(Dstp,i,j,0)+V(Dstp,i,j-1,0))+
                                     cff*(c1*V(rhs_vbar,i,j,0)+c2*VN(rvbar,i,j,0,kstp)-c3*VN(rvbar,i,j,0,ptsk)))*Dnew_avg*V(mskv,i,j,0);
            }
        }
        if(predictor_2d_step){                                                 /* :721-730 */
            FORBOX2D(xbxD,i,j) VN(rubar,i,j,0,krhs)=V(rhs_ubar,i,j,0);
            FORBOX2D(ybxD,i,j) VN(rvbar,i,j,0,krhs)=V(rhs_vbar,i,j,0);
        }
        o_fillpatch(g,&s->ubar,F(msku),BCV_UBAR,knew,1);                       /* :735-755 */
        o_fillpatch(g,&s->vbar,F(mskv),BCV_VBAR,knew,1);
        o_fillpatch(g,&s->zeta,F(mskr),BCV_ZETA,knew,1);
    }
    fab_free(&Drhs); fab_free(&fomn); fab_free(&Dnew); fab_free(&zwrk); fab_free(&gzeta); fab_free(&gzeta2); fab_free(&gzetaSA);
    fab_free(&Dstp); fab_free(&rhs_ubar); fab_free(&rhs_vbar); fab_free(&rhs_zeta); fab_free(&zeta_new);
    mf_free(&mf_DUon); mf_free(&mf_DVom);
}

/* TimeIntegration/REMORA_advance_2d_onestep.cpp:6-53 */
void o_advance_2d_onestep(OState *s,int my_iif)
{
    int nfast_counter=s->nfast+1;
    double dtfast_lev=s->dt/(double)s->prm.ndtfast;
    int first_2d_step=(my_iif==0);
    int next_indx1=0;
    advance_2d(s,dtfast_lev,1,first_2d_step,my_iif,&next_indx1);
    if(my_iif<nfast_counter-1) advance_2d(s,dtfast_lev,0,first_2d_step,my_iif,&next_indx1);
}

/* ------------------------------------------------------------------ advance_3d
 * TimeIntegration/REMORA_advance_3d.cpp:10-387 */
static void advance_3d(OState *s)
{
    const Geom *g=&s->geom;
    const int nrhs=0; int nnew=0; int iic=s->istep, ntfirst=0; int N=s->N; double dt_lev=s->dt;
    stretch_transform(s);                                                      /* :40 */
    Box bx=s->bx;
    Box gbx2=box_grow(bx,NGROW,NGROW,0), gbx21=box_grow(bx,NGROW,NGROW,NGROW-1), gbx1=box_grow(bx,NGROW-1,NGROW-1,0);
    Box xbx=box_nodal(bx,0), ybx=box_nodal(bx,1);
    MF mf_AK=mf_alloc(bx,0,0,1,1,NGROW,NGROW,0), mf_DC=mf_alloc(bx,0,0,0,1,NGROW,NGROW,NGROW-1), mf_Hzk=mf_alloc(bx,0,0,0,1,NGROW,NGROW,NGROW-1);
    Fab u=F(xvel_new), v=F(yvel_new), ru=F(ru), rv=F(rv), Hz=F(Hz), pm=F(pm), pn=F(pn);
    {
        Fab FC=fab_alloc(box_nodal(gbx2,2),1), BC=fab_alloc(box_nodal(gbx2,2),1), CF=fab_alloc(gbx21,1);
        double cff;
        if(iic==ntfirst) cff=0.25*dt_lev; else if(iic==ntfirst+1) cff=0.25*dt_lev*3.0/2.0; else cff=0.25*dt_lev*23.0/12.0;
        FORBOX(xbx,i,j,k){                                                     /* :113-117 */
            V(u,i,j,k)+=cff*(V(pm,i,j,0)+V(pm,i-1,j,0))*(V(pn,i,j,0)+V(pn,i-1,j,0))*VN(ru,i,j,k,nrhs);
            V(u,i,j,k)*=2.0/(V(Hz,i-1,j,k)+V(Hz,i,j,k));
        }
        FORBOX(ybx,i,j,k){                                                     /* :119-123 */
            V(v,i,j,k)+=cff*(V(pm,i,j,0)+V(pm,i,j-1,0))*(V(pn,i,j,0)+V(pn,i,j-1,0))*VN(rv,i,j,k,nrhs);
            V(v,i,j,k)*=2.0/(V(Hz,i,j-1,k)+V(Hz,i,j,k));
        }
        fab_setval_box(mf_DC.f,xbx,0.0); fab_setval_box(CF,xbx,0.0);
        o_vert_visc_3d(xbx,1,0,u,Hz,mf_Hzk.f,mf_AK.f,F(Akv),BC,mf_DC.f,FC,CF,nnew,N,dt_lev);
        fab_setval_box(mf_DC.f,ybx,0.0); fab_setval_box(CF,ybx,0.0);
        o_vert_visc_3d(ybx,0,1,v,Hz,mf_Hzk.f,mf_AK.f,F(Akv),BC,mf_DC.f,FC,CF,nnew,N,dt_lev);
        fab_setval_box(mf_DC.f,xbx,0.0); fab_setval_box(CF,xbx,0.0);
        o_vert_mean_3d(xbx,1,0,u,Hz,F(DU_avg1),mf_DC.f,CF,pn,F(msku),nnew,N);
        fab_setval_box(mf_DC.f,ybx,0.0); fab_setval_box(CF,ybx,0.0);
        o_vert_mean_3d(ybx,0,1,v,Hz,F(DV_avg1),mf_DC.f,CF,pm,F(mskv),nnew,N);
        fab_free(&FC); fab_free(&BC); fab_free(&CF);
    }
    { Fab none; memset(&none,0,sizeof none);                                   /* :154-155, mf_calc = xvel_old / yvel_old */
      o_physbc_calc(g,&s->xvel_new,F(msku),0,1,BCV_XVEL,0,F(xvel_old),none,none);
      o_physbc_calc(g,&s->yvel_new,F(mskv),0,1,BCV_YVEL,0,F(yvel_old),none,none); }
    {
        Fab FC=fab_alloc(gbx2,1);                                              /* :194-218 */
        fab_setval_box(FC,gbx2,0.0); fab_setval_box(mf_DC.f,box_grow(gbx2,0,0,1),0.0);
        o_update_massflux_3d(g,gbx2,1,0,u,F(ubar),F(Huon),Hz,pn,F(DU_avg1),F(DU_avg2),mf_DC.f,FC,F(msku),nnew);
        fab_setval_box(FC,gbx2,0.0); fab_setval_box(mf_DC.f,box_grow(gbx2,0,0,1),0.0);
        o_update_massflux_3d(g,gbx2,0,1,v,F(vbar),F(Hvom),Hz,pm,F(DV_avg1),F(DV_avg2),mf_DC.f,FC,F(mskv),nnew);
        fab_free(&FC);
    }
    MF mf_W=mf_alloc(bx,0,0,1,1,NGROW+1,NGROW+1,0); fab_setval(mf_W.f,0.0);    /* :230-294 */
    o_omega(gbx1,mf_W.f,F(Huon),F(Hvom),F(z_w),F(h),N);
    if(s->prm.gls.on){                                                         /* :296-303 */
        int nstp_g=iic%2, nnew_g=1-nstp_g;
        o_gls_corrector(g,bx,F(gls),F(tke),mf_W.f,F(Akv),F(Akt),F(Akk),F(Akp),F(Lscale),F(mskr),F(msku),F(mskv),Hz,
                        F(xvel_old),F(yvel_old),pm,pn,F(Huon),F(Hvom),F(z_w),F(sustr),F(svstr),F(bustr),F(bvstr),F(bvf),
                        nstp_g,nnew_g,N,dt_lev);
        for(int ic=0;ic<3;ic++){                                               /* REMORA_gls.cpp:905-914 */
            o_fillpatch(g,&s->tke,F(mskr),BCV_ZVEL,ic,1);
            o_fillpatch(g,&s->gls,F(mskr),BCV_ZVEL,ic,1);
        }
        for(int ic=0;ic<NCONS;ic++) o_fillpatch(g,&s->Akt,F(mskr),BCV_ZVEL,ic,1);
        o_fill_boundary(g,&s->Akv,0,1); o_fill_boundary(g,&s->Akp,0,1); o_fill_boundary(g,&s->Akk,0,1);
    }
    {
        Fab FC=fab_alloc(box_nodal(gbx2,2),1);                                 /* :306-348 */
        Box gbx=box_grow(bx,3,3,3);
        for(int n=0;n<NCONS;n++)
            o_rhs_t_3d(g,bx,gbx,fab_comp(F(cons_new),n),fab_comp(F(sstore),n),F(Huon),F(Hvom),Hz,pn,pm,mf_W.f,FC,
                       F(msku),F(mskv),nrhs,nnew,N,dt_lev);
        fab_free(&FC);
    }
    o_fillpatch_calc(g,&s->cons_new,F(mskr),BCV_CONS,0,NCONS,F(cons_old),F(msku),F(mskv));   /* :350, mf_calc = cons_old */
    {
        Box tbxp2=box_grow(bx,NGROW,NGROW,0), tbxp21=box_grow(bx,NGROW,NGROW,NGROW-1);
        /* reference allocates FC,BC on tbxp2 (cell) although vert_visc indexes k=N+1 on DC only; give them the k-face box */
        Fab FC=fab_alloc(box_nodal(tbxp2,2),1), BC=fab_alloc(box_nodal(tbxp2,2),1), CF=fab_alloc(tbxp21,1);
        for(int n=0;n<NCONS;n++)                                               /* :382-385 */
            o_vert_visc_3d(bx,0,0,fab_comp(F(cons_new),n),Hz,mf_Hzk.f,mf_AK.f,fab_comp(F(Akt),n),BC,mf_DC.f,FC,CF,nnew,N,dt_lev);
        fab_free(&FC); fab_free(&BC); fab_free(&CF);
    }
    mf_free(&mf_W); mf_free(&mf_AK); mf_free(&mf_DC); mf_free(&mf_Hzk);
}

/* TimeIntegration/REMORA_advance_3d_ml.cpp:6-71 */
void o_advance_3d_ml(OState *s)
{
    const Geom *g=&s->geom;
    o_fill_boundary(g,&s->xvel_new,0,1);                                       /* :11-12 */
    o_fill_boundary(g,&s->yvel_new,0,1);
    advance_3d(s);
    for(int n=0;n<3;n++){ o_fill_boundary(g,&s->ubar,n,1); o_fill_boundary(g,&s->vbar,n,1); }   /* :33-38 */
    o_fillpatch_calc(g,&s->cons_new,F(mskr),BCV_CONS,0,NCONS,F(cons_old),F(msku),F(mskv));   /* :45, mf_calc = cons_old */
    o_fill_boundary(g,&s->xvel_new,0,1);
    o_fill_boundary(g,&s->yvel_new,0,1);
    o_fillpatch(g,&s->zvel_new,F(mskr),BCV_ZVEL,0,1);                           /* :48 */
    { Fab c=F(cons_new), m=F(mskr); for(int n=0;n<NCONS;n++) FORBOX(s->bx,i,j,k) VN(c,i,j,k,n)*=V(m,i,j,0); }   /* :51-62 */
}

void o_begin_step(OState *s)                                                   /* REMORA_TimeStep.cpp:48-61 */
{
    s->t_old=s->t_new; s->t_new+=s->dt;
    swap_mf(&s->cons_old,&s->cons_new); swap_mf(&s->xvel_old,&s->xvel_new);
    swap_mf(&s->yvel_old,&s->yvel_new); swap_mf(&s->zvel_old,&s->zvel_new);
}
void o_end_step(OState *s){ s->istep++; }

void o_advance(OState *s)                                                      /* REMORA_Advance.cpp:7-35 */
{
    o_begin_step(s);
    o_setup_step(s);
    if(s->prm.use_barotropic){
        int nfast_counter=s->nfast+1;
        for(int my_iif=0;my_iif<nfast_counter;my_iif++) o_advance_2d_onestep(s,my_iif);
    }
    o_advance_3d_ml(s);
    o_end_step(s);
}

/* REMORA::update_mskp, Initialization/REMORA_make_new_level.cpp:617-655: the vorticity-point mask from the four rho-point masks
 * around the node, on the cell box grown by one: 1 with all four wet or exactly one dry, 2 with two dry cells side by side (a
 * straight coast), 0 otherwise (two dry cells across a diagonal, three or four dry) */
void o_update_mskp(OState *s)
{
    Fab mr=F(mskr), mp=F(mskp);
    for(int j=s->bx.lo[1]-1;j<=s->bx.hi[1]+1;j++) for(int i=s->bx.lo[0]-1;i<=s->bx.hi[0]+1;i++){
        int a=V(mr,i-1,j,0)>0.5, b=V(mr,i,j,0)>0.5, c=V(mr,i-1,j-1,0)>0.5, d=V(mr,i,j-1,0)>0.5;   /* :631 order of the tests */
        int a0=V(mr,i-1,j,0)<0.5, b0=V(mr,i,j,0)<0.5, c0=V(mr,i-1,j-1,0)<0.5, d0=V(mr,i,j-1,0)<0.5;
        double v;
        if(a&&b&&c&&d) v=1.0;
        else if(a0&&b&&c&&d) v=1.0;
        else if(a&&b0&&c&&d) v=1.0;
        else if(a&&b&&c0&&d) v=1.0;
        else if(a&&b&&c&&d0) v=1.0;
        else if(a&&b0&&c&&d0) v=2.0;
        else if(a0&&b&&c0&&d) v=2.0;
        else if(a&&b&&c0&&d0) v=2.0;
        else if(a0&&b0&&c&&d) v=2.0;
        else v=0.0;
        V(mp,i,j,0)=v;
    }
    /* Initialization/REMORA_init_from_netcdf.cpp:408-412: ghost cells of the four masks across periodic sides */
    o_fill_boundary(&s->geom,&s->mskr,0,1); o_fill_boundary(&s->geom,&s->msku,0,1);
    o_fill_boundary(&s->geom,&s->mskv,0,1); o_fill_boundary(&s->geom,&s->mskp,0,1);
}

int o_nfast(const OState *s){ return s->nfast; }
int o_istep(const OState *s){ return s->istep; }

int o_field(OState *s,const char *name,double **p,int lo[3],int hi[3],int *nc)
{
#define FLD(nm) if(!strcmp(name,#nm)){ Fab f=s->nm.f; *p=f.p; for(int d=0;d<3;d++){lo[d]=f.lo[d];hi[d]=f.hi[d];} *nc=f.nc; return 1; }
    FLD(cons_old) FLD(cons_new) FLD(xvel_old) FLD(xvel_new) FLD(yvel_old) FLD(yvel_new) FLD(zvel_old) FLD(zvel_new)
    FLD(Zt_avg1) FLD(h) FLD(ubar) FLD(vbar) FLD(ru) FLD(rv) FLD(ru2d) FLD(rv2d) FLD(mskr) FLD(msku) FLD(mskv) FLD(mskp)
    FLD(z_w) FLD(z_r) FLD(Hz) FLD(Huon) FLD(Hvom) FLD(Akv) FLD(Akt) FLD(visc2_p) FLD(visc2_r) FLD(diff2)
    FLD(rufrc) FLD(rvfrc) FLD(sustr) FLD(svstr) FLD(rdrag) FLD(bustr) FLD(bvstr) FLD(DU_avg1) FLD(DU_avg2) FLD(DV_avg1) FLD(DV_avg2)
    FLD(rubar) FLD(rvbar) FLD(rzeta) FLD(zeta) FLD(pm) FLD(pn) FLD(fcor) FLD(sstore) FLD(rhoS) FLD(rhoA) FLD(bvf)
    FLD(tke) FLD(gls) FLD(Lscale) FLD(Akk) FLD(Akp)
#undef FLD
    if(!strcmp(name,"weight1")){ *p=s->weight1; lo[0]=lo[1]=lo[2]=0; hi[0]=2*s->prm.ndtfast; hi[1]=hi[2]=0; *nc=1; return 1; }
    if(!strcmp(name,"weight2")){ *p=s->weight2; lo[0]=lo[1]=lo[2]=0; hi[0]=2*s->prm.ndtfast; hi[1]=hi[2]=0; *nc=1; return 1; }
    return 0;
}

/* IO/REMORA_Plotfile.cpp:185-235 */
void o_plot(OState *s,int with_scalar,double *out)
{
    Fab c=F(cons_new), u=F(xvel_new), v=F(yvel_new), w=F(zvel_new);
    long n3=(long)s->nx*s->ny*s->nz; long q=0; int nc=0;
    FORBOX(s->bx,i,j,k) out[nc*n3+(q++)]=VN(c,i,j,k,0);
    nc++; q=0;
    FORBOX(s->bx,i,j,k) out[nc*n3+(q++)]=VN(c,i,j,k,1);
    nc++; q=0;
    if(with_scalar){ FORBOX(s->bx,i,j,k) out[nc*n3+(q++)]=VN(c,i,j,k,2); nc++; q=0; }
    FORBOX(s->bx,i,j,k) out[nc*n3+(q++)]=0.5*(V(u,i,j,k)+V(u,i+1,j,k));
    nc++; q=0;
    FORBOX(s->bx,i,j,k) out[nc*n3+(q++)]=0.5*(V(v,i,j,k)+V(v,i,j+1,k));
    nc++; q=0;
    FORBOX(s->bx,i,j,k) out[nc*n3+(q++)]=0.5*(V(w,i,j,k)+V(w,i,j,k+1));
}

/* REMORA::estTimeStep, TimeIntegration/REMORA_ComputeTimestep.cpp:39-95: the cfl-limited dt from the cell-centred averages of the
 * face velocities (amrex::average_face_to_cellcenter, then ReduceMax :52-66); returns cfl / max, 1e20 if the max is not positive */
double o_est_timestep(OState *s,double cfl)
{
    const Params *P=&s->prm;
    Fab u=F(xvel_new), v=F(yvel_new), w=F(zvel_new);
    double dxi[3]={ s->nx/(P->prob_hi[0]-P->prob_lo[0]), s->ny/(P->prob_hi[1]-P->prob_lo[1]), s->nz/(P->prob_hi[2]-P->prob_lo[2]) };
    double m=-1.e100;
    FORBOX(s->bx,i,j,k){
        double cu=0.5*(V(u,i,j,k)+V(u,i+1,j,k)), cv=0.5*(V(v,i,j,k)+V(v,i,j+1,k)), cw=0.5*(V(w,i,j,k)+V(w,i,j,k+1));
        m=dmax(dmax(dmax(fabs(cu)*dxi[0],fabs(cv)*dxi[1]),fabs(cw)*dxi[2]),m);
    }
    return m>0.0 ? cfl/m : 1.e20;
}
/* REMORA::sum_integrated_quantities + volWgtSumMF, Source/REMORA_SumIQ.cpp:9-141: out = { scalar, kineng, volume, max_vel };
 * the sums are MultiFab::Dot(tmp, volume) accumulated serially in FAB order (k, j, i) */
void o_sum_iq(OState *s,double *out)
{
    Fab c=F(cons_new), u=F(xvel_new), v=F(yvel_new), Hz=F(Hz), pm=F(pm), pn=F(pn);
    double scalar=0.0, kineng=0.0, volume=0.0, kemax=-1.e300;
    FORBOX(s->bx,i,j,k){
        double ke=0.25*(V(u,i,j,k)*V(u,i,j,k)+V(u,i+1,j,k)*V(u,i+1,j,k)+V(v,i,j,k)*V(v,i,j,k)+V(v,i,j+1,k)*V(v,i,j+1,k));
        double vol=V(Hz,i,j,k)/(V(pm,i,j,0)*V(pn,i,j,0));
        if(kemax<ke) kemax=ke;
        scalar+=VN(c,i,j,k,2)*vol; kineng+=ke*vol; volume+=1.0*vol;
    }
    out[0]=scalar; out[1]=kineng; out[2]=volume; out[3]=sqrt(2.0*kemax);
}
