/*
 * oracle/leaves.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Pass-by-pass restatement of the Box/Array4-level leaf routines of REMORA's
 * Source/TimeIntegration.  Each function keeps the reference's pass structure and
 * per-expression operation order so that results agree with the gold plotfiles.
 * Build with -ffp-contract=off.
 */
#include "oracle.h"

/* ------------------------------------------------------------------ rho_eos
 * Source/TimeIntegration/REMORA_rho_eos.cpp:22-99 */
void o_rho_eos(const Geom *g,Box bx,Fab state,Fab rho,Fab rhoA,Fab rhoS,Fab bvf,Fab Hz,Fab z_w,Fab z_r,Fab h,Fab mskr,int N)
{
    const Params *P=g->prm;
    double R0=P->R0,S0=P->S0,T0=P->T0,Tcoef=P->Tcoef,Scoef=P->Scoef;
    FORBOX(bx,i,j,k) {                                                        /* :54-59 */
        V(rho,i,j,k)=(R0 - R0*Tcoef*(VN(state,i,j,k,0)-T0)
                         + R0*Scoef*(VN(state,i,j,k,1)-S0)
                         - 1000.0) * V(mskr,i,j,0);
    }
    double cff2=1.0/P->rho0;
    FORBOX2D(bx,i,j) {                                                        /* :69-86 */
        double cff0=V(rho,i,j,N)*V(Hz,i,j,N);
        V(rhoS,i,j,0)=0.5*cff0*V(Hz,i,j,N);
        V(rhoA,i,j,0)=cff0;
        for(int k=1;k<=N;k++){
            double cff1=V(rho,i,j,N-k)*V(Hz,i,j,N-k);
            V(rhoS,i,j,0)+=V(Hz,i,j,N-k)*(V(rhoA,i,j,0)+0.5*cff1);
            V(rhoA,i,j,0)+=cff1;
        }
        double cff11=1.0/(V(z_w,i,j,N+1)+VN(h,i,j,0,0));
        V(rhoA,i,j,0)*=cff2*cff11;
        V(rhoS,i,j,0)*=2.0*cff11*cff11*cff2;
    }
    double gorho0=P->g/P->rho0;
    Box bw=bx; bw.hi[2]+=1; bw.lo[2]+=1; bw.hi[2]-=1;                          /* :91-93 */
    FORBOX(bw,i,j,k) {
        V(bvf,i,j,k)=-gorho0*(V(rho,i,j,k)-V(rho,i,j,k-1))/(V(z_r,i,j,k)-V(z_r,i,j,k-1));
    }
}

/* ------------------------------------------------------------------ prsgrd
 * Source/TimeIntegration/REMORA_prsgrd.cpp:6-197 */
void o_prsgrd(const Geom *g,Box phi_bx,Box phi_gbx,Box utbx,Box vtbx,Fab ru,Fab rv,Fab pn,Fab pm,Fab rho,Fab FC,
              Fab Hz,Fab z_r,Fab z_w,Fab msku,Fab mskv,int nrhs,int N)
{
    (void)phi_gbx;
    const Params *Pm=g->prm;
    Box phi_bxD=box_slab(phi_bx,2,0);
    Box phi_ubx=box_nodal(phi_bx,0), phi_vbx=box_nodal(phi_bx,1);
    Box utbxD=box_slab(utbx,2,0), vtbxD=box_slab(vtbx,2,0);
    const double OneFifth=0.2, OneTwelfth=1.0/12.0, eps=1.0E-10;
    double GRho=Pm->g/Pm->rho0, GRho0=1000.0*GRho, HalfGRho=0.5*GRho;

    Fab P=fab_alloc(phi_bx,1), aux=fab_alloc(fab_box(z_r),1), dR=fab_alloc(phi_bx,1), dZ=fab_alloc(phi_bx,1),
        dRx=fab_alloc(phi_bx,1), dZx=fab_alloc(phi_bx,1);

    FORBOX(phi_bx,i,j,k) {                                                    /* :54-64 */
        if(k>=0&&k<N){ V(dR,i,j,k)=V(rho,i,j,k+1)-V(rho,i,j,k); V(dZ,i,j,k)=V(z_r,i,j,k+1)-V(z_r,i,j,k); }
        else { V(dR,i,j,N)=V(rho,i,j,N)-V(rho,i,j,N-1); V(dZ,i,j,N)=V(z_r,i,j,N)-V(z_r,i,j,N-1); }
    }
    FORBOX2D(phi_bxD,i,j) {                                                   /* :66-79 */
        for(int k=N;k>=0;k--){
            double cff= k>0 ? 2.0*V(dR,i,j,k)*V(dR,i,j,k-1) : 2.0*V(dR,i,j,k)*V(dR,i,j,k);
            if(cff>eps) V(dR,i,j,k)= k>0 ? cff/(V(dR,i,j,k)+V(dR,i,j,k-1)) : cff/(V(dR,i,j,k)+V(dR,i,j,k));
            else        V(dR,i,j,k)=0.0;
            V(dZ,i,j,k)= k>0 ? 2.0*V(dZ,i,j,k)*V(dZ,i,j,k-1)/(V(dZ,i,j,k)+V(dZ,i,j,k-1))
                             : 2.0*V(dZ,i,j,k)*V(dZ,i,j,k)/(V(dZ,i,j,k)+V(dZ,i,j,k));
        }
    }
    FORBOX2D(phi_bxD,i,j) {                                                   /* :81-99 */
        double cff1=1.0/(V(z_r,i,j,N)-V(z_r,i,j,N-1));
        double cff2=0.5*(V(rho,i,j,N)-V(rho,i,j,N-1))*(V(z_w,i,j,N+1)-V(z_r,i,j,N))*cff1;
        V(P,i,j,N)=GRho0*V(z_w,i,j,N+1)+GRho*(V(rho,i,j,N)+cff2)*(V(z_w,i,j,N+1)-V(z_r,i,j,N));
        for(int k=N-1;k>=0;k--){
            double rho_diff=V(rho,i,j,k+1)-V(rho,i,j,k)-OneTwelfth*(V(dR,i,j,k+1)+V(dR,i,j,k));
            double z_diff  =V(z_r,i,j,k+1)-V(z_r,i,j,k)-OneTwelfth*(V(dZ,i,j,k+1)+V(dZ,i,j,k));
            double rz_avg  =(V(rho,i,j,k+1)+V(rho,i,j,k))*(V(z_r,i,j,k+1)-V(z_r,i,j,k));
            V(P,i,j,k)=V(P,i,j,k+1)+HalfGRho*(rz_avg-
                         OneFifth*((V(dR,i,j,k+1)-V(dR,i,j,k))*z_diff-
                                   (V(dZ,i,j,k+1)-V(dZ,i,j,k))*rho_diff));
        }
    }
    FORBOX(phi_ubx,i,j,k) {                                                   /* :103-108 */
        V(FC,i,j,k)=(V(rho,i,j,k)-V(rho,i-1,j,k))*V(msku,i,j,0);
        V(aux,i,j,k)=(V(z_r,i,j,k)-V(z_r,i-1,j,k))*V(msku,i,j,0);
    }
    FORBOX2D(phi_bxD,i,j) {                                                   /* :112-131 */
        for(int k=N;k>=0;k--){
            double cff=2.0*V(aux,i,j,k)*V(aux,i+1,j,k);
            if(cff>eps){ double cff1=1.0/(V(aux,i+1,j,k)+V(aux,i,j,k)); V(dZx,i,j,k)=cff*cff1; }
            else V(dZx,i,j,k)=0.0;
            double cff1=2.0*V(FC,i,j,k)*V(FC,i+1,j,k);
            if(cff1>eps){ double cff2=1.0/(V(FC,i,j,k)+V(FC,i+1,j,k)); V(dRx,i,j,k)=cff1*cff2; }
            else V(dRx,i,j,k)=0.0;
        }
    }
    FORBOX2D(utbxD,i,j) {                                                     /* :134-149 */
        for(int k=N;k>=0;k--){
            double rho_diff=V(rho,i,j,k)-V(rho,i-1,j,k)-OneTwelfth*(V(dRx,i,j,k)+V(dRx,i-1,j,k));
            double z_r_diff=V(z_r,i,j,k)-V(z_r,i-1,j,k)-OneTwelfth*(V(dZx,i,j,k)+V(dZx,i-1,j,k));
            double Hz_avg=0.5*(V(Hz,i,j,k)+V(Hz,i-1,j,k));
            double on_u=2.0/(V(pn,i-1,j,0)+V(pn,i,j,0));
            VN(ru,i,j,k,nrhs)=on_u*Hz_avg*(
                V(P,i-1,j,k)-V(P,i,j,k)-HalfGRho*
                ((V(rho,i,j,k)+V(rho,i-1,j,k))*(V(z_r,i,j,k)-V(z_r,i-1,j,k))-
                 OneFifth*((V(dRx,i,j,k)-V(dRx,i-1,j,k))*z_r_diff-
                           (V(dZx,i,j,k)-V(dZx,i-1,j,k))*rho_diff)));
        }
    }
    FORBOX(phi_vbx,i,j,k) {                                                   /* :152-156 */
        V(FC,i,j,k)=(V(rho,i,j,k)-V(rho,i,j-1,k))*V(mskv,i,j,0);
        V(aux,i,j,k)=(V(z_r,i,j,k)-V(z_r,i,j-1,k))*V(mskv,i,j,0);
    }
    FORBOX2D(phi_bxD,i,j) {                                                   /* :160-178 */
        for(int k=N;k>=0;k--){
            double cff=2.0*V(aux,i,j,k)*V(aux,i,j+1,k);
            if(cff>eps){ double cff1=1.0/(V(aux,i,j+1,k)+V(aux,i,j,k)); V(dZx,i,j,k)=cff*cff1; }
            else V(dZx,i,j,k)=0.0;
            double cff1=2.0*V(FC,i,j,k)*V(FC,i,j+1,k);
            if(cff1>eps){ double cff2=1.0/(V(FC,i,j,k)+V(FC,i,j+1,k)); V(dRx,i,j,k)=cff1*cff2; }
            else V(dRx,i,j,k)=0.0;
        }
    }
    FORBOX2D(vtbxD,i,j) {                                                     /* :181-196 */
        for(int k=N;k>=0;k--){
            double rho_diff=V(rho,i,j,k)-V(rho,i,j-1,k)-OneTwelfth*(V(dRx,i,j,k)+V(dRx,i,j-1,k));
            double z_r_diff=V(z_r,i,j,k)-V(z_r,i,j-1,k)-OneTwelfth*(V(dZx,i,j,k)+V(dZx,i,j-1,k));
            double Hz_avg=0.5*(V(Hz,i,j,k)+V(Hz,i,j-1,k));
            double om_v=2.0/(V(pm,i,j-1,0)+V(pm,i,j,0));
            VN(rv,i,j,k,nrhs)=om_v*Hz_avg*(
                V(P,i,j-1,k)-V(P,i,j,k)-HalfGRho*
                ((V(rho,i,j,k)+V(rho,i,j-1,k))*(V(z_r,i,j,k)-V(z_r,i,j-1,k))-
                 OneFifth*((V(dRx,i,j,k)-V(dRx,i,j-1,k))*z_r_diff-
                           (V(dZx,i,j,k)-V(dZx,i,j-1,k))*rho_diff)));
        }
    }
    fab_free(&P); fab_free(&aux); fab_free(&dR); fab_free(&dZ); fab_free(&dRx); fab_free(&dZx);
}

/* ------------------------------------------------------------------ omega (W)
 * Source/TimeIntegration/REMORA_prestep_t_advection.cpp:97-125 == REMORA_advance_3d.cpp:272-293 */
void o_omega(Box gbx1,Fab W,Fab Huon,Fab Hvom,Fab z_w,Fab h,int N)
{
    Box b=box_slab(gbx1,2,0);
    FORBOX2D(b,i,j) {
        V(W,i,j,0)=0.0;
        for(int k=1;k<=N+1;k++)
            V(W,i,j,k)=V(W,i,j,k-1)-(V(Huon,i+1,j,k-1)-V(Huon,i,j,k-1))-(V(Hvom,i,j+1,k-1)-V(Hvom,i,j,k-1));
    }
    FORBOX2D(b,i,j) {
        double wrk=V(W,i,j,N+1)/(V(z_w,i,j,N+1)+VN(h,i,j,0,0));
        for(int k=1;k<=N;k++) V(W,i,j,k)=V(W,i,j,k)-wrk*(V(z_w,i,j,k)+VN(h,i,j,0,0));
        V(W,i,j,N+1)=0.0;
    }
}

/* ------------------------------------------------------------------ prestep_t_advection
 * Source/TimeIntegration/REMORA_prestep_t_advection.cpp:10-361 */
void o_prestep_t_advection(const Geom *g,Box tbx,Box gbx,Fab tempold,Fab tempcache,Fab Hz,Fab Huon,Fab Hvom,Fab W,Fab DC,Fab FC,
              Fab tempstore,Fab z_w,Fab h,Fab pm,Fab pn,Fab msku,Fab mskv,int iic,int ntfirst,int nrhs,int N,double dt_lev)
{
    const Params *P=g->prm;
    Box tbxp1=box_grow(tbx,NGROW-1,NGROW-1,0), tbxp2=box_grow(tbx,NGROW,NGROW,0);
    Fab FX=fab_alloc(tbxp2,1), FE=fab_alloc(tbxp2,1), curv=fab_alloc(tbxp2,1), grad=fab_alloc(tbxp2,1);
    fab_setval(FX,0.0); fab_setval(FE,0.0); fab_setval(curv,0.0); fab_setval(grad,0.0);        /* :54-63 */
    Box ubx=box_nodal(tbx,0), vbx=box_nodal(tbx,1);
    Box utbxp1=box_nodal(tbxp1,0), vtbxp1=box_nodal(tbxp1,1);
    Box gbx1=box_and(box_grow(tbx,NGROW-1,NGROW-1,0),gbx);                                    /* :82-85 */

    o_omega(gbx1,W,Huon,Hvom,z_w,h,N);                                                        /* :97-125 */

    if(P->flat_bathymetry){                                                                   /* :147-157 */
        Box tb=fab_box(tempold);
        FORBOX(tbxp1,i,j,k)
            V(FX,i,j,k)= box_contains(tb,i-1,j,k) ? V(Huon,i,j,k)*0.5*(V(tempold,i-1,j,k)+V(tempold,i,j,k)) : 1e34;
        FORBOX(tbxp1,i,j,k)
            V(FE,i,j,k)= box_contains(tb,i,j-1,k) ? V(Hvom,i,j,k)*0.5*(V(tempold,i,j-1,k)+V(tempold,i,j,k)) : 1e34;
    } else {
        FORBOX(utbxp1,i,j,k) V(FX,i,j,k)=(VN(tempold,i,j,k,nrhs)-VN(tempold,i-1,j,k,nrhs))*V(msku,i,j,0);   /* :161-165 */
        /* :167-180 boundary-slab copies are dead code (guarded by !slab.ok()) */
        FORBOX(vtbxp1,i,j,k) V(FE,i,j,k)=(VN(tempold,i,j,k,nrhs)-VN(tempold,i,j-1,k,nrhs))*V(mskv,i,j,0);   /* :182-186 */
        double cffa=1.0/6.0, cffb=1.0/3.0;
        if(!P->tracer_c4){                                                                    /* :205-220 */
            FORBOX(tbxp1,i,j,k) V(curv,i,j,k)=-V(FX,i,j,k)+V(FX,i+1,j,k);
            FORBOX(tbxp1,i,j,k){
                double max_Huon=dmax(V(Huon,i,j,k),0.0), min_Huon=dmin(V(Huon,i,j,k),0.0);
                V(FX,i,j,k)=V(Huon,i,j,k)*0.5*(V(tempold,i,j,k)+V(tempold,i-1,j,k))-
                    cffa*(V(curv,i,j,k)*min_Huon+V(curv,i-1,j,k)*max_Huon);
            }
        } else {                                                                              /* :222-234 */
            FORBOX(tbxp1,i,j,k) V(grad,i,j,k)=0.5*(V(FX,i,j,k)+V(FX,i+1,j,k));
            FORBOX(ubx,i,j,k)
                V(FX,i,j,k)=V(Huon,i,j,k)*0.5*(V(tempold,i,j,k)+V(tempold,i-1,j,k)-
                                               cffb*(V(grad,i,j,k)-V(grad,i-1,j,k)));
        }
        if(!P->tracer_c4){                                                                    /* :239-253 */
            FORBOX(tbxp1,i,j,k) V(curv,i,j,k)=-V(FE,i,j,k)+V(FE,i,j+1,k);
            FORBOX(tbxp1,i,j,k){
                double max_Hvom=dmax(V(Hvom,i,j,k),0.0), min_Hvom=dmin(V(Hvom,i,j,k),0.0);
                V(FE,i,j,k)=V(Hvom,i,j,k)*0.5*(V(tempold,i,j,k)+V(tempold,i,j-1,k))-
                    cffa*(V(curv,i,j,k)*min_Hvom+V(curv,i,j-1,k)*max_Hvom);
            }
        } else {                                                                              /* :255-266 */
            FORBOX(tbxp1,i,j,k) V(grad,i,j,k)=0.5*(V(FE,i,j,k)+V(FE,i,j+1,k));
            FORBOX(vbx,i,j,k)
                V(FE,i,j,k)=V(Hvom,i,j,k)*0.5*(V(tempold,i,j,k)+V(tempold,i,j-1,k)-
                                               cffb*(V(grad,i,j,k)-V(grad,i,j-1,k)));
        }
    }

    double cff1,cff2,cff, GammaT=1.0/6.0;                                                     /* :277-290 */
    if(iic==ntfirst){ cff=0.5*dt_lev; cff1=1.0; cff2=0.0; }
    else { cff=(1.0-GammaT)*dt_lev; cff1=0.5+GammaT; cff2=0.5-GammaT; }

    FORBOX(tbx,i,j,k)                                                                         /* :292-298 */
        V(tempstore,i,j,k)=V(Hz,i,j,k)*(cff1*V(tempold,i,j,k)+cff2*V(tempcache,i,j,k))-
                           cff*V(pm,i,j,0)*V(pn,i,j,0)*(V(FX,i+1,j,k)-V(FX,i,j,k)+
                                                        V(FE,i,j+1,k)-V(FE,i,j,k));
    Box wbx=box_nodal(tbx,2);
    FORBOX(wbx,i,j,k){                                                                        /* :304-332 */
        double c1=0.5, c2=7.0/12.0, c3=1.0/12.0;
        if(k>=2&&k<=N-1)
            V(FC,i,j,k)=(c2*(VN(tempold,i,j,k-1,nrhs)+VN(tempold,i,j,k,nrhs))
                        -c3*(VN(tempold,i,j,k-2,nrhs)+VN(tempold,i,j,k+1,nrhs)))*(V(W,i,j,k));
        else if(k==N+1) V(FC,i,j,N+1)=0.0;
        else if(k==N)
            V(FC,i,j,N)=(c2*VN(tempold,i,j,N-1,nrhs)+c1*VN(tempold,i,j,N,nrhs)-c3*VN(tempold,i,j,N-2,nrhs))*V(W,i,j,N);
        else if(k==1)
            V(FC,i,j,1)=(c2*VN(tempold,i,j,1,nrhs)+c1*VN(tempold,i,j,0,nrhs)-c3*VN(tempold,i,j,2,nrhs))*V(W,i,j,1);
        else if(k==0) V(FC,i,j,0)=0.0;
    }
    FORBOX(tbxp1,i,j,k)                                                                       /* :334-349 */
        V(DC,i,j,k)=1.0/(V(Hz,i,j,k)-
                    cff*V(pm,i,j,0)*V(pn,i,j,0)*
                    (V(Huon,i+1,j,k)-V(Huon,i,j,k)+
                     V(Hvom,i,j+1,k)-V(Hvom,i,j,k)+
                     (V(W,i,j,k+1)-V(W,i,j,k))));
    FORBOX(tbx,i,j,k){                                                                        /* :351-359 */
        double c1=cff*V(pm,i,j,0)*V(pn,i,j,0);
        double c4=V(FC,i,j,k+1)-V(FC,i,j,k);
        V(tempstore,i,j,k)=V(DC,i,j,k)*(V(tempstore,i,j,k)-c1*c4);
    }
    fab_free(&FX); fab_free(&FE); fab_free(&curv); fab_free(&grad);
}

/* ------------------------------------------------------------------ prestep_diffusion
 * Source/TimeIntegration/REMORA_prestep_diffusion.cpp:19-165 */
void o_prestep_diffusion(const Geom *g,Box vel_bx,Box gbx,int ioff,int joff,Fab vel,Fab vel_old,Fab rvel,Fab Hz,Fab Akv,
              Fab DC,Fab FC,Fab sstr,Fab bstr,Fab z_r,Fab pm,Fab pn,int iic,int ntfirst,int nnew,int nstp,int nrhs,int N,
              double lambda,double dt_lev)
{
    (void)g;
    Box gbxvel=box_and(vel_bx,gbx);
    double oml_dt=dt_lev*(1.0-lambda);
    FORBOX(vel_bx,i,j,k){                                                                     /* :56-77 */
        if(k+1<=N&&k>=0){
            double cff=1.0/(V(z_r,i,j,k+1)+V(z_r,i-ioff,j-joff,k+1)
                           -V(z_r,i,j,k)-V(z_r,i-ioff,j-joff,k));
            V(FC,i,j,k)=oml_dt*cff*(VN(vel_old,i,j,k+1,nstp)-VN(vel_old,i,j,k,nstp))*
                                   (V(Akv,i,j,k)+V(Akv,i-ioff,j-joff,k));
        } else if(k==-1) V(FC,i,j,-1)=dt_lev*V(bstr,i,j,0);
        else if(k==N)    V(FC,i,j,N)=dt_lev*V(sstr,i,j,0);
        V(DC,i,j,k)=0.25*dt_lev*(V(pm,i,j,0)+V(pm,i-ioff,j-joff,0))
                               *(V(pn,i,j,0)+V(pn,i-ioff,j-joff,0));
    }
    FORBOX(gbxvel,i,j,k){                                                                     /* :79-164 */
        double cff3=dt_lev*(1.0-lambda), cff1=0.0, cff2=0.0, cff4;
        int indx=0;
        if(iic==ntfirst){
            if(k+1<=N&&k>=1){
                cff1=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff2=V(FC,i,j,k)-V(FC,i,j,k-1);
                VN(vel,i,j,k,nnew)=cff1+cff2;
            } else if(k==0){
                cff1=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff2=V(FC,i,j,k)-dt_lev*V(bstr,i,j,0);
                VN(vel,i,j,k,nnew)=cff1+cff2;
            } else if(k==N){
                cff1=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff2=dt_lev*V(sstr,i,j,0)-V(FC,i,j,k-1);
                VN(vel,i,j,k,nnew)=cff1+cff2;
            }
        } else if(iic==ntfirst+1){
            if(k<N&&k>0){
                cff1=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff2=V(FC,i,j,k)-V(FC,i,j,k-1);
            } else if(k==0){
                cff1=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff2=V(FC,i,j,k)-dt_lev*V(bstr,i,j,0);
            } else if(k==N){
                cff1=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff2=dt_lev*V(sstr,i,j,0)-V(FC,i,j,k-1);
            }
            cff3=0.5*V(DC,i,j,k);
            if(ioff==0&&joff==0){
                VN(vel,i,j,k,nnew)=cff1+cff2;
            } else {
                indx=nrhs?0:1;
                double r_swap=VN(rvel,i,j,k,indx);
                VN(rvel,i,j,k,indx)=VN(rvel,i,j,k,nrhs);
                VN(rvel,i,j,k,nrhs)=r_swap;
                VN(vel,i,j,k,nnew)=cff1-cff3*VN(rvel,i,j,k,indx)+cff2;
            }
        } else {
            cff1=5.0/12.0; cff2=16.0/12.0;
            if(k==0){
                cff3=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff4=V(FC,i,j,k)-dt_lev*V(bstr,i,j,0);
            } else if(k==N){
                cff3=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff4=dt_lev*V(sstr,i,j,0)-V(FC,i,j,k-1);
            } else {
                cff3=VN(vel_old,i,j,k,nstp)*0.5*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
                cff4=V(FC,i,j,k)-V(FC,i,j,k-1);
            }
            if(ioff==0&&joff==0){
                VN(vel,i,j,k,nnew)=cff3+cff4;
            } else {
                indx=nrhs?0:1;
                double r_swap=VN(rvel,i,j,k,indx);
                VN(rvel,i,j,k,indx)=VN(rvel,i,j,k,nrhs);
                VN(rvel,i,j,k,nrhs)=r_swap;
                VN(vel,i,j,k,nnew)=cff3+V(DC,i,j,k)*(cff1*VN(rvel,i,j,k,nrhs)-
                                                     cff2*VN(rvel,i,j,k,indx))+cff4;
                VN(rvel,i,j,k,nrhs)=0.0;
            }
        }
    }
}

/* ------------------------------------------------------------------ t3dmix
 * Source/TimeIntegration/REMORA_t3dmix.cpp:6-58 */
void o_t3dmix(const Geom *g,Box bx,Fab state,Fab state_rhs,Fab diff2,Fab Hz,Fab pm,Fab pn,Fab msku,Fab mskv,double dt_lev,int ncomp)
{
    (void)g;
    Box xbx=box_nodal(bx,0), ybx=box_nodal(bx,1);
    Fab FX=fab_alloc(xbx,ncomp), FE=fab_alloc(ybx,ncomp);
    for(int n=0;n<ncomp;n++) FORBOX(xbx,i,j,k){
        double pmon_u=(V(pm,i-1,j,0)+V(pm,i,j,0))/(V(pn,i-1,j,0)+V(pn,i,j,0));
        double cff=0.25*(V(diff2,i,j,n)+V(diff2,i-1,j,n))*pmon_u;   /* reference writes diff2(i,j,n): 3-index Array4 access, i.e. k=n of component 0 (:34) */
        VN(FX,i,j,k,n)=cff*(V(Hz,i,j,k)+V(Hz,i-1,j,k))*(VN(state_rhs,i,j,k,n)-VN(state_rhs,i-1,j,k,n));
        VN(FX,i,j,k,n)*=V(msku,i,j,0);
    }
    for(int n=0;n<ncomp;n++) FORBOX(ybx,i,j,k){
        double pnom_v=(V(pn,i,j-1,0)+V(pn,i,j,0))/(V(pm,i,j-1,0)+V(pm,i,j,0));
        double cff=0.25*(V(diff2,i,j,n)+V(diff2,i,j-1,n))*pnom_v;   /* :43, same quirk */
        VN(FE,i,j,k,n)=cff*(V(Hz,i,j,k)+V(Hz,i,j-1,k))*(VN(state_rhs,i,j,k,n)-VN(state_rhs,i,j-1,k,n));
        VN(FE,i,j,k,n)*=V(mskv,i,j,0);
    }
    for(int n=0;n<ncomp;n++) FORBOX(bx,i,j,k){
        double cff=dt_lev*V(pm,i,j,0)*V(pn,i,j,0);
        VN(state,i,j,k,n)+=cff*((VN(FX,i+1,j,k,n)-VN(FX,i,j,k,n))
                               +(VN(FE,i,j+1,k,n)-VN(FE,i,j,k,n)));
    }
    fab_free(&FX); fab_free(&FE);
}

/* ------------------------------------------------------------------ coriolis
 * Source/TimeIntegration/REMORA_coriolis.cpp:10-40 */
void o_coriolis(Box xbx,Box ybx,Fab uold,Fab vold,Fab ru,Fab rv,Fab Hz,Fab fomn,int nrhs,int nr)
{
    FORBOX(xbx,i,j,k){
        double UFx_i  =0.5*V(Hz,i  ,j,k)*V(fomn,i  ,j,0)*(VN(vold,i  ,j,k,nrhs)+VN(vold,i  ,j+1,k,nrhs));
        double UFx_im1=0.5*V(Hz,i-1,j,k)*V(fomn,i-1,j,0)*(VN(vold,i-1,j,k,nrhs)+VN(vold,i-1,j+1,k,nrhs));
        VN(ru,i,j,k,nr)+=0.5*(UFx_i+UFx_im1);
    }
    FORBOX(ybx,i,j,k){
        double VFe_j  =0.5*V(Hz,i,j  ,k)*V(fomn,i,j  ,0)*(VN(uold,i,j  ,k,nrhs)+VN(uold,i+1,j  ,k,nrhs));
        double VFe_jm1=0.5*V(Hz,i,j-1,k)*V(fomn,i,j-1,0)*(VN(uold,i,j-1,k,nrhs)+VN(uold,i+1,j-1,k,nrhs));
        VN(rv,i,j,k,nr)-=0.5*(VFe_j+VFe_jm1);
    }
}

/* ------------------------------------------------------------------ rhs_uv_3d
 * Source/TimeIntegration/REMORA_rhs_uv_3d.cpp:31-385 */
void o_rhs_uv_3d(const Geom *g,Box xbx,Box ybx,Fab uold,Fab vold,Fab ru,Fab rv,Fab rufrc,Fab rvfrc,Fab sustr,Fab svstr,
              Fab bustr,Fab bvstr,Fab Huon,Fab Hvom,Fab pm,Fab pn,Fab W,Fab FC,int nrhs,int N)
{
    const Params *P=g->prm;
    int dlox=g->domain.lo[0], dhix=g->domain.hi[0], dloy=g->domain.lo[1], dhiy=g->domain.hi[1];
    int perx=g->per[0], pery=g->per[1];
    Box bUFx=box_growlo(xbx,0,1), bUFe=box_growhi(xbx,1,1), bVFe=box_growlo(ybx,1,1), bVFx=box_growhi(ybx,0,1);
    Fab UFx=fab_alloc(bUFx,1), UFe=fab_alloc(bUFe,1), VFe=fab_alloc(bVFe,1), VFx=fab_alloc(bVFx,1);
    fab_setval(UFx,0.0); fab_setval(UFe,0.0); fab_setval(VFe,0.0); fab_setval(VFx,0.0);
    const double Gadv=-0.25;

    if(!P->uv_c2){
        FORBOX(bUFx,i,j,k){                                                                   /* :96-121 */
            double cff1=VN(uold,i,j,k,nrhs)+VN(uold,i+1,j,k,nrhs);
            double uxx_i  =VN(uold,i-1,j,k,nrhs)-2.0*VN(uold,i  ,j,k,nrhs)+VN(uold,i+1,j,k,nrhs);
            double uxx_ip1=VN(uold,i  ,j,k,nrhs)-2.0*VN(uold,i+1,j,k,nrhs)+VN(uold,i+2,j,k,nrhs);
            double Huxx_i  =V(Huon,i-1,j,k)-2.0*V(Huon,i  ,j,k)+V(Huon,i+1,j,k);
            double Huxx_ip1=V(Huon,i  ,j,k)-2.0*V(Huon,i+1,j,k)+V(Huon,i+2,j,k);
            if(i==dlox&&!perx){ uxx_i=uxx_ip1; Huxx_i=Huxx_ip1; }
            else if(i==dhix&&!perx){ uxx_ip1=uxx_i; Huxx_ip1=Huxx_i; }
            double cff=(cff1>0.0)?uxx_i:uxx_ip1;
            double Huon_avg=(V(Huon,i,j,k)+V(Huon,i+1,j,k));
            V(UFx,i,j,k)=0.25*(cff1+Gadv*cff)*(Huon_avg+0.5*Gadv*(Huxx_i+Huxx_ip1));
        }
        FORBOX(bUFe,i,j,k){                                                                   /* :126-147 */
            double cff1=VN(uold,i,j,k,nrhs)+VN(uold,i,j-1,k,nrhs);
            double cff2=V(Hvom,i,j,k)+V(Hvom,i-1,j,k);
            double uee_jm1=VN(uold,i,j-2,k,nrhs)-2.0*VN(uold,i,j-1,k,nrhs)+VN(uold,i,j,k,nrhs);
            double uee_j  =VN(uold,i,j-1,k,nrhs)-2.0*VN(uold,i,j  ,k,nrhs)+VN(uold,i,j+1,k,nrhs);
            if(j==dloy&&!pery) uee_jm1=uee_j;
            else if(j==dhiy+1&&!pery) uee_j=uee_jm1;
            double cff=(cff2>0.0)?uee_jm1:uee_j;
            double Hvxx_i  =V(Hvom,i-1,j,k)-2.0*V(Hvom,i  ,j,k)+V(Hvom,i+1,j,k);
            double Hvxx_im1=V(Hvom,i-2,j,k)-2.0*V(Hvom,i-1,j,k)+V(Hvom,i  ,j,k);
            V(UFe,i,j,k)=0.25*(cff1+Gadv*cff)*(cff2+Gadv*0.5*(Hvxx_i+Hvxx_im1));
        }
    } else {
        FORBOX(bUFx,i,j,k) V(UFx,i,j,k)=0.25*(VN(uold,i,j,k,nrhs)+VN(uold,i+1,j,k,nrhs))*(V(Huon,i,j,k)+V(Huon,i+1,j,k));
        FORBOX(bUFe,i,j,k) V(UFe,i,j,k)=0.25*(VN(uold,i,j-1,k,nrhs)+VN(uold,i,j,k,nrhs))*(V(Hvom,i-1,j,k)+V(Hvom,i,j,k));
    }
    FORBOX(xbx,i,j,k)                                                                         /* :162-165 */
        VN(ru,i,j,k,nrhs)-=((V(UFx,i,j,k)-V(UFx,i-1,j,k))+(V(UFe,i,j+1,k)-V(UFe,i,j,k)));

    Box xw=box_nodal(xbx,2);
    if(!P->uv_c2){
        FORBOX(xw,i,j,k){                                                                     /* :168-198 */
            double cff1=9.0/16.0, cff2=1.0/16.0;
            if(k>1&&k<=N-1){
                V(FC,i,j,k)=(cff1*(VN(uold,i,j,k-1,nrhs)+VN(uold,i,j,k  ,nrhs))
                            -cff2*(VN(uold,i,j,k-2,nrhs)+VN(uold,i,j,k+1,nrhs)))*
                            (cff1*(V(W,i  ,j,k)+V(W,i-1,j,k))
                            -cff2*(V(W,i+1,j,k)+V(W,i-2,j,k)));
            } else {
                V(FC,i,j,N+1)=0.0;
                V(FC,i,j,N)=(cff1*(VN(uold,i,j,N-1,nrhs)+VN(uold,i,j,N,nrhs))
                            -cff2*(VN(uold,i,j,N-2,nrhs)+VN(uold,i,j,N,nrhs)))*
                            (cff1*(V(W,i  ,j,N)+V(W,i-1,j,N))
                            -cff2*(V(W,i+1,j,N)+V(W,i-2,j,N)));
                V(FC,i,j,1)=(cff1*(VN(uold,i,j,0,nrhs)+VN(uold,i,j,1,nrhs))
                            -cff2*(VN(uold,i,j,0,nrhs)+VN(uold,i,j,2,nrhs)))*
                            (cff1*(V(W,i  ,j,1)+V(W,i-1,j,1))
                            -cff2*(V(W,i+1,j,1)+V(W,i-2,j,1)));
                V(FC,i,j,0)=0.0;
            }
        }
    } else {
        FORBOX(xw,i,j,k){
            if(k>0&&k<=N) V(FC,i,j,k)=0.25*(VN(uold,i,j,k-1,nrhs)+VN(uold,i,j,k,nrhs))*(V(W,i,j,k)+V(W,i-1,j,k));
            else { V(FC,i,j,N+1)=0.0; V(FC,i,j,0)=0.0; }
        }
    }
    FORBOX(xbx,i,j,k){ double cff=V(FC,i,j,k+1)-V(FC,i,j,k); VN(ru,i,j,k,nrhs)-=cff; }        /* :211-216 */
    { Box b=box_slab(xbx,2,0);
      FORBOX2D(b,i,j) for(int k=0;k<=N;k++){                                                  /* :221-236 */
        V(rufrc,i,j,0)+=VN(ru,i,j,k,nrhs);
        double om_u=2.0/(V(pm,i-1,j,0)+V(pm,i,j,0));
        double on_u=2.0/(V(pn,i-1,j,0)+V(pn,i,j,0));
        double cff=om_u*on_u;
        double cff1=(k==N)? V(sustr,i,j,0)*cff:0.0;
        double cff2=(k==0)?-V(bustr,i,j,0)*cff:0.0;
        V(rufrc,i,j,0)+=cff1+cff2;
      } }

    if(!P->uv_c2){
        FORBOX(bVFx,i,j,k){                                                                   /* :254-276 */
            double cff1=VN(vold,i,j,k,nrhs)+VN(vold,i-1,j,k,nrhs);
            double cff2=V(Huon,i,j,k)+V(Huon,i,j-1,k);
            double vxx_im1=VN(vold,i-2,j,k,nrhs)-2.0*VN(vold,i-1,j,k,nrhs)+VN(vold,i  ,j,k,nrhs);
            double vxx_i  =VN(vold,i-1,j,k,nrhs)-2.0*VN(vold,i  ,j,k,nrhs)+VN(vold,i+1,j,k,nrhs);
            if(i==dlox&&!perx) vxx_im1=vxx_i;
            else if(i==dhix+1&&!perx) vxx_i=vxx_im1;
            double cff=(cff2>0.0)?vxx_im1:vxx_i;
            double Huee_j  =V(Huon,i,j-1,k)-2.0*V(Huon,i,j  ,k)+V(Huon,i,j+1,k);
            double Huee_jm1=V(Huon,i,j-2,k)-2.0*V(Huon,i,j-1,k)+V(Huon,i,j  ,k);
            V(VFx,i,j,k)=0.25*(cff1+Gadv*cff)*(cff2+Gadv*0.5*(Huee_j+Huee_jm1));
        }
        FORBOX(bVFe,i,j,k){                                                                   /* :279-301 */
            double cff1=VN(vold,i,j,k,nrhs)+VN(vold,i,j+1,k,nrhs);
            double vee_j  =VN(vold,i,j-1,k,nrhs)-2.0*VN(vold,i,j  ,k,nrhs)+VN(vold,i,j+1,k,nrhs);
            double vee_jp1=VN(vold,i,j  ,k,nrhs)-2.0*VN(vold,i,j+1,k,nrhs)+VN(vold,i,j+2,k,nrhs);
            double Hvee_j  =V(Hvom,i,j-1,k)-2.0*V(Hvom,i,j  ,k)+V(Hvom,i,j+1,k);
            double Hvee_jp1=V(Hvom,i,j  ,k)-2.0*V(Hvom,i,j+1,k)+V(Hvom,i,j+2,k);
            if(j==dloy&&!pery){ vee_j=vee_jp1; Hvee_j=Hvee_jp1; }
            else if(j==dhiy&&!pery){ vee_jp1=vee_j; Hvee_jp1=Hvee_j; }
            double cff=(cff1>0.0)?vee_j:vee_jp1;
            V(VFe,i,j,k)=0.25*(cff1+Gadv*cff)*(V(Hvom,i,j,k)+V(Hvom,i,j+1,k)+0.5*Gadv*(Hvee_j+Hvee_jp1));
        }
    } else {
        FORBOX(bVFx,i,j,k) V(VFx,i,j,k)=0.25*(VN(vold,i-1,j,k,nrhs)+VN(vold,i,j,k,nrhs))*(V(Huon,i,j-1,k)+V(Huon,i,j,k));
        FORBOX(bVFe,i,j,k) V(VFe,i,j,k)=0.25*(VN(vold,i,j,k,nrhs)+VN(vold,i,j+1,k,nrhs))*(V(Hvom,i,j,k)+V(Hvom,i,j+1,k));
    }
    FORBOX(ybx,i,j,k)                                                                         /* :313-316 */
        VN(rv,i,j,k,nrhs)-=((V(VFx,i+1,j,k)-V(VFx,i,j,k))+(V(VFe,i,j,k)-V(VFe,i,j-1,k)));

    Box yw=box_nodal(ybx,2);
    if(!P->uv_c2){
        FORBOX(yw,i,j,k){                                                                     /* :319-345 */
            double cff1=9.0/16.0, cff2=1.0/16.0;
            if(k>1&&k<=N-1){
                V(FC,i,j,k)=(cff1*(VN(vold,i,j,k-1,nrhs)+VN(vold,i,j,k  ,nrhs))
                            -cff2*(VN(vold,i,j,k-2,nrhs)+VN(vold,i,j,k+1,nrhs)))*
                            (cff1*(V(W,i,j  ,k)+V(W,i,j-1,k))
                            -cff2*(V(W,i,j+1,k)+V(W,i,j-2,k)));
            } else {
                V(FC,i,j,N+1)=0.0;
                V(FC,i,j,N)=(cff1*(VN(vold,i,j,N-1,nrhs)+VN(vold,i,j,N,nrhs))
                            -cff2*(VN(vold,i,j,N-2,nrhs)+VN(vold,i,j,N,nrhs)))*
                            (cff1*(V(W,i,j  ,N)+V(W,i,j-1,N))
                            -cff2*(V(W,i,j+1,N)+V(W,i,j-2,N)));
                V(FC,i,j,1)=(cff1*(VN(vold,i,j,0,nrhs)+VN(vold,i,j,1,nrhs))
                            -cff2*(VN(vold,i,j,0,nrhs)+VN(vold,i,j,2,nrhs)))*
                            (cff1*(V(W,i,j  ,1)+V(W,i,j-1,1))
                            -cff2*(V(W,i,j+1,1)+V(W,i,j-2,1)));
                V(FC,i,j,0)=0.0;
            }
        }
    } else {
        FORBOX(yw,i,j,k){
            if(k>0&&k<=N) V(FC,i,j,k)=0.25*(VN(vold,i,j,k-1,nrhs)+VN(vold,i,j,k,nrhs))*(V(W,i,j,k)+V(W,i,j-1,k));
            else { V(FC,i,j,N+1)=0.0; V(FC,i,j,0)=0.0; }
        }
    }
    FORBOX(ybx,i,j,k){ double cff=V(FC,i,j,k+1)-V(FC,i,j,k); VN(rv,i,j,k,nrhs)-=cff; }        /* :359-364 */
    { Box b=box_slab(ybx,2,0);
      FORBOX2D(b,i,j) for(int k=0;k<=N;k++){                                                  /* :369-384 */
        V(rvfrc,i,j,0)+=VN(rv,i,j,k,nrhs);
        double om_v=2.0/(V(pm,i,j-1,0)+V(pm,i,j,0));
        double on_v=2.0/(V(pn,i,j-1,0)+V(pn,i,j,0));
        double cff=om_v*on_v;
        double cff1=(k==N)? V(svstr,i,j,0)*cff:0.0;
        double cff2=(k==0)?-V(bvstr,i,j,0)*cff:0.0;
        V(rvfrc,i,j,0)+=cff1+cff2;
      } }
    fab_free(&UFx); fab_free(&UFe); fab_free(&VFe); fab_free(&VFx);
}

/* ------------------------------------------------------------------ uv3dmix
 * Source/TimeIntegration/REMORA_uv3dmix.cpp:6-149 */
void o_uv3dmix(Box xbx,Box ybx,Fab u,Fab v,Fab uold,Fab vold,Fab rufrc,Fab rvfrc,Fab visc2_p,Fab visc2_r,Fab Hz,
              Fab pm,Fab pn,Fab mskp,int nrhs,int nnew,double dt_lev)
{
    Box bUFx=box_growlo(xbx,0,1), bUFe=box_growhi(xbx,1,1), bVFe=box_growlo(ybx,1,1), bVFx=box_growhi(ybx,0,1);
    Fab UFx=fab_alloc(bUFx,1), UFe=fab_alloc(bUFe,1), VFe=fab_alloc(bVFe,1), VFx=fab_alloc(bVFx,1);
    fab_setval(UFx,0.0); fab_setval(UFe,0.0); fab_setval(VFe,0.0); fab_setval(VFx,0.0);
    int N=xbx.hi[2]-ybx.lo[2];                                                                /* :36 */

    FORBOX(bUFx,i,j,k){                                                                       /* :48-59 */
        double cff=0.5*V(Hz,i,j,k)*(V(pm,i,j,0)/V(pn,i,j,0)*
                   ((V(pn,i  ,j,0)+V(pn,i+1,j,0))*VN(uold,i+1,j,k,nrhs)-
                    (V(pn,i-1,j,0)+V(pn,i  ,j,0))*VN(uold,i  ,j,k,nrhs))-
                   V(pn,i,j,0)/V(pm,i,j,0)*
                   ((V(pm,i,j  ,0)+V(pm,i,j+1,0))*VN(vold,i,j+1,k,nrhs)-
                    (V(pm,i,j-1,0)+V(pm,i,j  ,0))*VN(vold,i,j  ,k,nrhs)));
        double on_r=1.0/V(pn,i,j,0);
        V(UFx,i,j,k)=on_r*on_r*V(visc2_r,i,j,0)*cff;
    }
    FORBOX(bUFe,i,j,k){                                                                       /* :61-78 */
        double pmon_p=(V(pm,i-1,j-1,0)+V(pm,i-1,j,0)+V(pm,i,j-1,0)+V(pm,i,j,0))/
                      (V(pn,i-1,j-1,0)+V(pn,i-1,j,0)+V(pn,i,j-1,0)+V(pn,i,j,0));
        double pnom_p=(V(pn,i-1,j-1,0)+V(pn,i-1,j,0)+V(pn,i,j-1,0)+V(pn,i,j,0))/
                      (V(pm,i-1,j-1,0)+V(pm,i-1,j,0)+V(pm,i,j-1,0)+V(pm,i,j,0));
        double cff=V(mskp,i,j,0)*0.125*
                   (V(Hz,i-1,j,k)+V(Hz,i,j,k)+V(Hz,i-1,j-1,k)+V(Hz,i,j-1,k))*
                   (pmon_p*
                    ((V(pn,i  ,j-1,0)+V(pn,i  ,j,0))*VN(vold,i  ,j,k,nrhs)-
                     (V(pn,i-1,j-1,0)+V(pn,i-1,j,0))*VN(vold,i-1,j,k,nrhs))+
                    pnom_p*
                    ((V(pm,i-1,j  ,0)+V(pm,i,j  ,0))*VN(uold,i,j  ,k,nrhs)-
                     (V(pm,i-1,j-1,0)+V(pm,i,j-1,0))*VN(uold,i,j-1,k,nrhs)));
        double om_p=4.0/(V(pm,i-1,j-1,0)+V(pm,i-1,j,0)+V(pm,i,j-1,0)+V(pm,i,j,0));
        V(UFe,i,j,k)=om_p*om_p*V(visc2_p,i,j,0)*cff;
    }
    { Box b=box_slab(xbx,2,0);
      FORBOX2D(b,i,j) for(int k=0;k<=N;k++){                                                  /* :80-90 */
        double cff=dt_lev*0.25*(V(pm,i-1,j,0)+V(pm,i,j,0))*(V(pn,i-1,j,0)+V(pn,i,j,0));
        double cff1=0.5*(V(pn,i-1,j,0)+V(pn,i,j,0))*(V(UFx,i,j  ,k)-V(UFx,i-1,j,k));
        double cff2=0.5*(V(pm,i-1,j,0)+V(pm,i,j,0))*(V(UFe,i,j+1,k)-V(UFe,i  ,j,k));
        double cff3=cff*(cff1+cff2);
        VN(u,i,j,k,nnew)=VN(u,i,j,k,nnew)+cff3;
        V(rufrc,i,j,0)+=cff1+cff2;
      } }
    FORBOX(bVFe,i,j,k){                                                                       /* :103-115 */
        double cff=0.5*V(Hz,i,j,k)*(V(pm,i,j,0)/V(pn,i,j,0)*
                   ((V(pn,i  ,j,0)+V(pn,i+1,j,0))*VN(uold,i+1,j,k,nrhs)-
                    (V(pn,i-1,j,0)+V(pn,i  ,j,0))*VN(uold,i  ,j,k,nrhs))-
                   V(pn,i,j,0)/V(pm,i,j,0)*
                   ((V(pm,i,j  ,0)+V(pm,i,j+1,0))*VN(vold,i,j+1,k,nrhs)-
                    (V(pm,i,j-1,0)+V(pm,i,j  ,0))*VN(vold,i,j  ,k,nrhs)));
        double om_r=1.0/V(pm,i,j,0);
        V(VFe,i,j,k)=om_r*om_r*V(visc2_r,i,j,0)*cff;
    }
    FORBOX(bVFx,i,j,k){                                                                       /* :117-134 */
        double pmon_p=(V(pm,i-1,j-1,0)+V(pm,i-1,j,0)+V(pm,i,j-1,0)+V(pm,i,j,0))/
                      (V(pn,i-1,j-1,0)+V(pn,i-1,j,0)+V(pn,i,j-1,0)+V(pn,i,j,0));
        double pnom_p=(V(pn,i-1,j-1,0)+V(pn,i-1,j,0)+V(pn,i,j-1,0)+V(pn,i,j,0))/
                      (V(pm,i-1,j-1,0)+V(pm,i-1,j,0)+V(pm,i,j-1,0)+V(pm,i,j,0));
        double cff=V(mskp,i,j,0)*0.125*(V(Hz,i-1,j,k)+V(Hz,i,j,k)+
                                       V(Hz,i-1,j-1,k)+V(Hz,i,j-1,k))*
                   (pmon_p*
                    ((V(pn,i  ,j-1,0)+V(pn,i  ,j,0))*VN(vold,i  ,j,k,nrhs)-
                     (V(pn,i-1,j-1,0)+V(pn,i-1,j,0))*VN(vold,i-1,j,k,nrhs))+
                    pnom_p*
                    ((V(pm,i-1,j  ,0)+V(pm,i,j  ,0))*VN(uold,i,j  ,k,nrhs)-
                     (V(pm,i-1,j-1,0)+V(pm,i,j-1,0))*VN(uold,i,j-1,k,nrhs)));
        double on_p=4.0/(V(pn,i-1,j-1,0)+V(pn,i-1,j,0)+V(pn,i,j-1,0)+V(pn,i,j,0));
        V(VFx,i,j,k)=on_p*on_p*V(visc2_p,i,j,0)*cff;
    }
    { Box b=box_slab(ybx,2,0);
      FORBOX2D(b,i,j) for(int k=0;k<=N;k++){                                                  /* :136-147 */
        double cff=dt_lev*0.25*(V(pm,i,j,0)+V(pm,i,j-1,0))*(V(pn,i,j,0)+V(pn,i,j-1,0));
        double cff1=0.5*(V(pn,i,j-1,0)+V(pn,i,j,0))*(V(VFx,i+1,j,k)-V(VFx,i,j  ,k));
        double cff2=0.5*(V(pm,i,j-1,0)+V(pm,i,j,0))*(V(VFe,i  ,j,k)-V(VFe,i,j-1,k));
        double cff3=cff*(cff1-cff2);
        VN(v,i,j,k,nnew)=VN(v,i,j,k,nnew)+cff3;
        V(rvfrc,i,j,0)+=cff1-cff2;
      } }
    fab_free(&UFx); fab_free(&UFe); fab_free(&VFe); fab_free(&VFx);
}

/* ------------------------------------------------------------------ vert_visc_3d
 * Source/TimeIntegration/REMORA_vert_visc_3d.cpp:10-107 */
void o_vert_visc_3d(Box phi_bx,int ioff,int joff,Fab phi,Fab Hz,Fab Hzk,Fab AK,Fab Akv,Fab BC,Fab DC,Fab FC,Fab CF,
              int nnew,int N,double dt_lev)
{
    FORBOX(phi_bx,i,j,k) V(Hzk,i,j,k)=0.5*(V(Hz,i-ioff,j-joff,k)+V(Hz,i,j,k));                /* :28-31 */
    Box wb=box_nodal(phi_bx,2);
    FORBOX(wb,i,j,k) V(AK,i,j,k)=0.5*(V(Akv,i-ioff,j-joff,k)+V(Akv,i,j,k));                   /* :36-39 */
    const double sixth=1.0/6.0, third=1.0/3.0;
    Box b2=box_slab(phi_bx,2,0);
    FORBOX2D(b2,i,j){                                                                         /* :51-75 */
        V(DC,i,j,0)=0.0; V(FC,i,j,0)=0.0; V(CF,i,j,0)=0.0;
        for(int k=1;k<=N;k++){
            double oHzkm1=1.0/V(Hzk,i,j,k-1);
            double oHz=1.0/V(Hzk,i,j,k);
            V(FC,i,j,k)=sixth*V(Hzk,i,j,k-1)-dt_lev*V(AK,i,j,k-1)/V(Hzk,i,j,k-1);
            V(CF,i,j,k)=sixth*V(Hzk,i,j,k  )-dt_lev*V(AK,i,j,k+1)/V(Hzk,i,j,k  );
            V(BC,i,j,k)=third*(V(Hzk,i,j,k-1)+V(Hzk,i,j,k))+dt_lev*V(AK,i,j,k)*(oHzkm1+oHz);
            double cff=1.0/(V(BC,i,j,k)-V(FC,i,j,k)*V(CF,i,j,k-1));
            V(CF,i,j,k)*=cff;
            V(DC,i,j,k)=cff*(VN(phi,i,j,k,nnew)-VN(phi,i,j,k-1,nnew)-V(FC,i,j,k)*V(DC,i,j,k-1));
        }
    }
    FORBOX2D(b2,i,j){                                                                         /* :81-90 */
        V(DC,i,j,N+1)=0.0;
        for(int k=N;k>=1;k--) V(DC,i,j,k)-=V(CF,i,j,k)*V(DC,i,j,k+1);
    }
    FORBOX(wb,i,j,k) V(DC,i,j,k)*=V(AK,i,j,k);                                                /* :96-99 */
    FORBOX(phi_bx,i,j,k){                                                                     /* :101-106 */
        double oHz=1.0/V(Hzk,i,j,k);
        double cff=dt_lev*oHz*(V(DC,i,j,k+1)-V(DC,i,j,k));
        V(phi,i,j,k)+=cff;
    }
}

/* ------------------------------------------------------------------ vert_mean_3d
 * Source/TimeIntegration/REMORA_vert_mean_3d.cpp:10-47 */
void o_vert_mean_3d(Box phi_bx,int ioff,int joff,Fab phi,Fab Hz,Fab Dphi_avg1,Fab DC,Fab CF,Fab dxlen,Fab msk,int nnew,int N)
{
    Box b2=box_slab(phi_bx,2,0);
    FORBOX2D(b2,i,j){
        double Hzk_on_face=0.5*(V(Hz,i-ioff,j-joff,0)+V(Hz,i,j,0));
        V(CF,i,j,-1)=Hzk_on_face;
        V(DC,i,j,-1)=VN(phi,i,j,0,nnew)*Hzk_on_face;
        for(int k=1;k<=N;k++){
            Hzk_on_face=0.5*(V(Hz,i-ioff,j-joff,k)+V(Hz,i,j,k));
            V(CF,i,j,-1)+=Hzk_on_face;
            V(DC,i,j,-1)+=VN(phi,i,j,k,nnew)*Hzk_on_face;
        }
    }
    FORBOX2D(b2,i,j){
        double on_u_or_om_v=2.0/(V(dxlen,i-ioff,j-joff,0)+V(dxlen,i,j,0));
        double cff1=1.0/(V(CF,i,j,-1)*(on_u_or_om_v));
        V(DC,i,j,-1)=(V(DC,i,j,-1)*(on_u_or_om_v)-V(Dphi_avg1,i,j,0))*cff1;
    }
    FORBOX(phi_bx,i,j,k){
        V(phi,i,j,k)-=V(DC,i,j,-1);
        V(phi,i,j,k)*=V(msk,i,j,0);
    }
}

/* ------------------------------------------------------------------ update_massflux_3d
 * Source/TimeIntegration/REMORA_update_massflux_3d.cpp:24-122 */
void o_update_massflux_3d(const Geom *g,Box bx,int ioff,int joff,Fab phi,Fab phibar,Fab Hphi,Fab Hz,Fab pm_or_pn,
              Fab Dphi_avg1,Fab Dphi_avg2,Fab DC,Fab FC,Fab msk,int nnew)
{
    int dlox=g->domain.lo[0], dhix=g->domain.hi[0], dloy=g->domain.lo[1], dhiy=g->domain.hi[1];
    int perx=g->per[0], pery=g->per[1];
    int N=g->domain.hi[2]-g->domain.lo[2];
    Box bxD=box_slab(bx,2,0);
    Fab CF=fab_alloc(bxD,1); fab_setval(CF,0.0);
    FORBOX(bx,i,j,k){                                                                         /* :70-75 */
        double om_v_or_on_u=2.0/(V(pm_or_pn,i,j,0)+V(pm_or_pn,i-ioff,j-joff,0));
        V(DC,i,j,k)=0.5*om_v_or_on_u*(V(Hz,i,j,k)+V(Hz,i-ioff,j-joff,k));
    }
    FORBOX2D(bxD,i,j){                                                                        /* :77-109 */
        for(int k=0;k<=N;k++){
            V(DC,i,j,-1)+=V(DC,i,j,k);
            V(CF,i,j,0)+=V(DC,i,j,k)*VN(phi,i,j,k,nnew);
        }
        V(DC,i,j,-1)=1.0/V(DC,i,j,-1);
        V(CF,i,j,0)=V(DC,i,j,-1)*(V(CF,i,j,0)-V(Dphi_avg1,i,j,0));
        for(int k=0;k<=N;k++){
            if(i==dlox-joff&&!perx){ VN(phi,i,j,k,nnew)-=V(CF,i,j,0); VN(phi,i,j,k,nnew)*=V(msk,i,j,0); }
            else if(i==dhix+1&&!perx){ VN(phi,i,j,k,nnew)-=V(CF,i,j,0); VN(phi,i,j,k,nnew)*=V(msk,i,j,0); }
            else if(j==dloy-ioff&&!pery){ VN(phi,i,j,k,nnew)-=V(CF,i,j,0); VN(phi,i,j,k,nnew)*=V(msk,i,j,0); }
            else if(j==dhiy+1&&!pery){ VN(phi,i,j,k,nnew)-=V(CF,i,j,0); VN(phi,i,j,k,nnew)*=V(msk,i,j,0); }
        }
        for(int k=0;k<=N;k++){
            V(Hphi,i,j,k)=0.5*(V(Hphi,i,j,k)+VN(phi,i,j,k,nnew)*V(DC,i,j,k));
            V(FC,i,j,0)+=V(Hphi,i,j,k);
        }
        V(FC,i,j,0)=V(DC,i,j,-1)*(V(FC,i,j,0)-V(Dphi_avg2,i,j,0));
    }
    FORBOX(bx,i,j,k) V(Hphi,i,j,k)-=V(DC,i,j,k)*V(FC,i,j,0);                                  /* :111-114 */
    FORBOX2D(bxD,i,j){                                                                        /* :116-120 */
        VN(phibar,i,j,0,0)=V(DC,i,j,-1)*V(Dphi_avg1,i,j,0);
        VN(phibar,i,j,0,1)=VN(phibar,i,j,0,0);
    }
    fab_free(&CF);
}

/* ------------------------------------------------------------------ rhs_t_3d
 * Source/TimeIntegration/REMORA_rhs_t_3d.cpp:27-335 */
void o_rhs_t_3d(const Geom *g,Box bx,Box gbx,Fab t,Fab sstore,Fab Huon,Fab Hvom,Fab Hz,Fab pn,Fab pm,Fab W,Fab FC,
              Fab msku,Fab mskv,int nrhs,int nnew,int N,double dt_lev)
{
    (void)gbx;
    const Params *P=g->prm;
    Box tbxp2=box_grow(bx,NGROW,NGROW,0), tbxp1x=box_grow(bx,NGROW-1,0,0), tbxp1y=box_grow(bx,0,NGROW-1,0);
    Box utbxp1=box_nodal(tbxp1x,0), vtbxp1=box_nodal(tbxp1y,1), ubx=box_nodal(bx,0), vbx=box_nodal(bx,1);
    Fab grad=fab_alloc(tbxp2,1), curv=fab_alloc(tbxp2,1), FX=fab_alloc(tbxp2,1), FE=fab_alloc(tbxp2,1);
    fab_setval(grad,0.0); fab_setval(curv,0.0); fab_setval(FX,0.0); fab_setval(FE,0.0);
    double cffa=1.0/6.0, cffb=1.0/3.0;
    /* sign of the curvature / gradient correction: flat-bathymetry branch flips it (:124-125,:138-139 vs :161-162,:175-176) */
    int flat=P->flat_bathymetry;

    FORBOX(utbxp1,i,j,k) V(FX,i,j,k)=(VN(sstore,i,j,k,nrhs)-VN(sstore,i-1,j,k,nrhs))*V(msku,i,j,0);     /* :87-90 */
    if(!P->tracer_c4){
        FORBOX(tbxp1x,i,j,k) V(curv,i,j,k)=-V(FX,i,j,k)+V(FX,i+1,j,k);
        FORBOX(ubx,i,j,k){
            double max_Huon=dmax(V(Huon,i,j,k),0.0), min_Huon=dmin(V(Huon,i,j,k),0.0);
            if(flat) V(FX,i,j,k)=V(Huon,i,j,k)*0.5*(V(sstore,i,j,k)+V(sstore,i-1,j,k))+
                         cffa*(V(curv,i,j,k)*min_Huon+V(curv,i-1,j,k)*max_Huon);
            else     V(FX,i,j,k)=V(Huon,i,j,k)*0.5*(V(sstore,i,j,k)+V(sstore,i-1,j,k))-
                         cffa*(V(curv,i,j,k)*min_Huon+V(curv,i-1,j,k)*max_Huon);
        }
    } else {
        FORBOX(tbxp1x,i,j,k) V(grad,i,j,k)=0.5*(V(FX,i,j,k)+V(FX,i+1,j,k));
        FORBOX(ubx,i,j,k){
            if(flat) V(FX,i,j,k)=V(Huon,i,j,k)*0.5*(V(sstore,i,j,k)+V(sstore,i-1,j,k))+
                         cffb*(V(grad,i,j,k)+V(grad,i-1,j,k));
            else     V(FX,i,j,k)=V(Huon,i,j,k)*0.5*(V(sstore,i,j,k)+V(sstore,i-1,j,k)-
                         cffb*(V(grad,i,j,k)-V(grad,i-1,j,k)));
        }
    }
    FORBOX(vtbxp1,i,j,k) V(FE,i,j,k)=(VN(sstore,i,j,k,nrhs)-VN(sstore,i,j-1,k,nrhs))*V(mskv,i,j,0);     /* :184-187 */
    if(!P->tracer_c4){
        FORBOX(tbxp1y,i,j,k) V(curv,i,j,k)=-V(FE,i,j,k)+V(FE,i,j+1,k);
        FORBOX(vbx,i,j,k){
            double max_Hvom=dmax(V(Hvom,i,j,k),0.0), min_Hvom=dmin(V(Hvom,i,j,k),0.0);
            if(flat) V(FE,i,j,k)=V(Hvom,i,j,k)*0.5*(V(sstore,i,j,k)+V(sstore,i,j-1,k))+
                         cffa*(V(curv,i,j,k)*min_Hvom+V(curv,i,j-1,k)*max_Hvom);
            else     V(FE,i,j,k)=V(Hvom,i,j,k)*0.5*(V(sstore,i,j,k)+V(sstore,i,j-1,k))-
                         cffa*(V(curv,i,j,k)*min_Hvom+V(curv,i,j-1,k)*max_Hvom);
        }
    } else {
        FORBOX(tbxp1y,i,j,k) V(grad,i,j,k)=0.5*(V(FE,i,j,k)+V(FE,i,j+1,k));
        FORBOX(vbx,i,j,k){
            if(flat) V(FE,i,j,k)=V(Hvom,i,j,k)*0.5*(V(sstore,i,j,k)+V(sstore,i,j-1,k))+
                         cffb*(V(grad,i,j,k)+V(grad,i,j-1,k));
            else     V(FE,i,j,k)=V(Hvom,i,j,k)*0.5*(V(sstore,i,j,k)+V(sstore,i,j-1,k)-
                         cffb*(V(grad,i,j,k)-V(grad,i,j-1,k)));
        }
    }
    FORBOX(bx,i,j,k){                                                                         /* :278-290 */
        double cff=dt_lev*V(pm,i,j,0)*V(pn,i,j,0);
        double cff1=cff*(V(FX,i+1,j,k)-V(FX,i,j,k));
        double cff2=cff*(V(FE,i,j+1,k)-V(FE,i,j,k));
        double cff3=cff1+cff2;
        VN(t,i,j,k,nnew)-=cff3;
    }
    Box wb=box_nodal(bx,2);
    FORBOX(wb,i,j,k){                                                                         /* :301-326 */
        double cff1=0.5, cff2=7.0/12.0, cff3=1.0/12.0;
        if(k>=2&&k<=N-1)
            V(FC,i,j,k)=(cff2*(V(sstore,i,j,k-1)+V(sstore,i,j,k))
                        -cff3*(V(sstore,i,j,k-2)+V(sstore,i,j,k+1)))*(V(W,i,j,k));
        else if(k==N+1) V(FC,i,j,N+1)=0.0;
        else if(k==N)
            V(FC,i,j,N)=(cff2*V(sstore,i,j,N-1)+cff1*V(sstore,i,j,N)-cff3*V(sstore,i,j,N-2))*(V(W,i,j,N));
        else if(k==1)
            V(FC,i,j,1)=(cff2*V(sstore,i,j,1)+cff1*V(sstore,i,j,0)-cff3*V(sstore,i,j,2))*(V(W,i,j,1));
        else if(k==0) V(FC,i,j,0)=0.0;
    }
    FORBOX(bx,i,j,k){                                                                         /* :328-334 */
        double cff1=dt_lev*V(pm,i,j,0)*V(pn,i,j,0);
        double cff4=V(FC,i,j,k+1)-V(FC,i,j,k);
        V(t,i,j,k)=(V(t,i,j,k)-cff1*cff4)/V(Hz,i,j,k);
    }
    fab_free(&grad); fab_free(&curv); fab_free(&FX); fab_free(&FE);
}

/* ------------------------------------------------------------------ rhs_uv_2d
 * Source/TimeIntegration/REMORA_rhs_uv_2d.cpp:20-238 */
void o_rhs_uv_2d(const Geom *g,Box xbx,Box ybx,Fab ubar,Fab vbar,Fab rhs_ubar,Fab rhs_vbar,Fab DUon,Fab DVom,int krhs)
{
    const Params *P=g->prm;
    int dlox=g->domain.lo[0], dhix=g->domain.hi[0], dloy=g->domain.lo[1], dhiy=g->domain.hi[1];
    int perx=g->per[0], pery=g->per[1];
    Box bUFx=box_growlo(xbx,0,1), bUFe=box_growhi(xbx,1,1), bVFe=box_growlo(ybx,1,1), bVFx=box_growhi(ybx,0,1);
    Fab UFx=fab_alloc(bUFx,1), UFe=fab_alloc(bUFe,1), VFe=fab_alloc(bVFe,1), VFx=fab_alloc(bVFx,1);
    fab_setval(UFx,0.0); fab_setval(UFe,0.0); fab_setval(VFe,0.0); fab_setval(VFx,0.0);
    if(!P->uv_c2){
        FORBOX2D(bUFx,i,j){                                                                   /* :80-102 */
            double uxx_i  =VN(ubar,i-1,j,0,krhs)-2.0*VN(ubar,i  ,j,0,krhs)+VN(ubar,i+1,j,0,krhs);
            double uxx_ip1=VN(ubar,i  ,j,0,krhs)-2.0*VN(ubar,i+1,j,0,krhs)+VN(ubar,i+2,j,0,krhs);
            double Huxx_i  =V(DUon,i-1,j,0)-2.0*V(DUon,i  ,j,0)+V(DUon,i+1,j,0);
            double Huxx_ip1=V(DUon,i  ,j,0)-2.0*V(DUon,i+1,j,0)+V(DUon,i+2,j,0);
            if(i==dlox&&!perx){ uxx_i=uxx_ip1; Huxx_i=Huxx_ip1; }
            else if(i==dhix&&!perx){ uxx_ip1=uxx_i; Huxx_ip1=Huxx_i; }
            double cff=1.0/6.0;
            double ubar_avg=VN(ubar,i,j,0,krhs)+VN(ubar,i+1,j,0,krhs);
            V(UFx,i,j,0)=0.25*(ubar_avg-cff*(uxx_i+uxx_ip1))*(V(DUon,i,j,0)+V(DUon,i+1,j,0)-cff*(Huxx_i+Huxx_ip1));
        }
        FORBOX2D(bUFe,i,j){                                                                   /* :107-129 */
            double uee_j  =VN(ubar,i,j-1,0,krhs)-2.0*VN(ubar,i,j  ,0,krhs)+VN(ubar,i,j+1,0,krhs);
            double uee_jm1=VN(ubar,i,j-2,0,krhs)-2.0*VN(ubar,i,j-1,0,krhs)+VN(ubar,i,j  ,0,krhs);
            double Hvxx_i  =V(DVom,i-1,j,0)-2.0*V(DVom,i  ,j,0)+V(DVom,i+1,j,0);
            double Hvxx_im1=V(DVom,i-2,j,0)-2.0*V(DVom,i-1,j,0)+V(DVom,i  ,j,0);
            if(j==dloy&&!pery) uee_jm1=uee_j;
            else if(j==dhiy+1&&!pery) uee_j=uee_jm1;
            double cff=1.0/6.0;
            double cff1=VN(ubar,i,j,0,krhs)+VN(ubar,i,j-1,0,krhs);
            double cff2=V(DVom,i,j,0)+V(DVom,i-1,j,0);
            V(UFe,i,j,0)=0.25*(cff1-(uee_j+uee_jm1)*cff)*(cff2-cff*(Hvxx_i+Hvxx_im1));
        }
    } else {
        FORBOX2D(bUFx,i,j) V(UFx,i,j,0)=0.25*(V(DUon,i,j,0)+V(DUon,i+1,j,0))*(VN(ubar,i,j,0,krhs)+VN(ubar,i+1,j,0,krhs));
        FORBOX2D(bUFe,i,j) V(UFe,i,j,0)=0.25*(V(DVom,i,j,0)+V(DVom,i-1,j,0))*(VN(ubar,i,j,0,krhs)+VN(ubar,i,j-1,0,krhs));
    }
    { Box b=box_slab(xbx,2,0);
      FORBOX2D(b,i,j){                                                                        /* :147-153 */
        double cff1=V(UFx,i,j,0)-V(UFx,i-1,j,0);
        double cff2=V(UFe,i,j+1,0)-V(UFe,i,j,0);
        V(rhs_ubar,i,j,0)-=(cff1+cff2);
      } }
    if(!P->uv_c2){
        FORBOX2D(bVFx,i,j){                                                                   /* :171-191 */
            double cff=1.0/6.0;
            double vxx_i  =VN(vbar,i-1,j,0,krhs)-2.0*VN(vbar,i  ,j,0,krhs)+VN(vbar,i+1,j,0,krhs);
            double vxx_im1=VN(vbar,i-2,j,0,krhs)-2.0*VN(vbar,i-1,j,0,krhs)+VN(vbar,i  ,j,0,krhs);
            double Huee_j  =V(DUon,i,j-1,0)-2.0*V(DUon,i,j  ,0)+V(DUon,i,j+1,0);
            double Huee_jm1=V(DUon,i,j-2,0)-2.0*V(DUon,i,j-1,0)+V(DUon,i,j  ,0);
            if(i==dlox&&!perx) vxx_im1=vxx_i;
            else if(i==dhix+1&&!perx) vxx_i=vxx_im1;
            double cff1=VN(vbar,i,j,0,krhs)+VN(vbar,i-1,j,0,krhs);
            double cff2=V(DUon,i,j,0)+V(DUon,i,j-1,0);
            V(VFx,i,j,0)=0.25*(cff1-(vxx_i+vxx_im1)*cff)*(cff2-cff*(Huee_j+Huee_jm1));
        }
        FORBOX2D(bVFe,i,j){                                                                   /* :193-216 */
            double vee_j  =VN(vbar,i,j-1,0,krhs)-2.0*VN(vbar,i,j  ,0,krhs)+VN(vbar,i,j+1,0,krhs);
            double vee_jp1=VN(vbar,i,j  ,0,krhs)-2.0*VN(vbar,i,j+1,0,krhs)+VN(vbar,i,j+2,0,krhs);
            double Hvee_j  =V(DVom,i,j-1,0)-2.0*V(DVom,i,j  ,0)+V(DVom,i,j+1,0);
            double Hvee_jp1=V(DVom,i,j  ,0)-2.0*V(DVom,i,j+1,0)+V(DVom,i,j+2,0);
            if(j==dloy&&!pery){ vee_j=vee_jp1; Hvee_j=Hvee_jp1; }
            else if(j==dhiy&&!pery){ vee_jp1=vee_j; Hvee_jp1=Hvee_j; }
            double cff=1.0/6.0;
            double cff1=VN(vbar,i,j,0,krhs)+VN(vbar,i,j+1,0,krhs);
            V(VFe,i,j,0)=0.25*(cff1-(vee_j+vee_jp1)*cff)*(V(DVom,i,j,0)+V(DVom,i,j+1,0)-
                                                         cff*(Hvee_j+Hvee_jp1));
        }
    } else {
        FORBOX2D(bVFx,i,j) V(VFx,i,j,0)=0.25*(V(DUon,i,j,0)+V(DUon,i,j-1,0))*(VN(vbar,i,j,0,krhs)+VN(vbar,i-1,j,0,krhs));
        FORBOX2D(bVFe,i,j) V(VFe,i,j,0)=0.25*(V(DVom,i,j,0)+V(DVom,i,j+1,0))*(VN(vbar,i,j,0,krhs)+VN(vbar,i,j+1,0,krhs));
    }
    { Box b=box_slab(ybx,2,0);
      FORBOX2D(b,i,j){                                                                        /* :230-236 */
        double cff1=V(VFx,i+1,j,0)-V(VFx,i,j,0);
        double cff2=V(VFe,i,j,0)-V(VFe,i,j-1,0);
        V(rhs_vbar,i,j,0)-=(cff1+cff2);
      } }
    fab_free(&UFx); fab_free(&UFe); fab_free(&VFe); fab_free(&VFx);
}

/* ------------------------------------------------------------------ set_weights
 * Source/TimeIntegration/REMORA_set_weights.cpp:10-171 */
int o_set_weights(int ndtfast,double *weight1,double *weight2)
{
    double gamma,scale,wsum,shift,cff;
    double Falpha=2.0,Fbeta=4.0,Fgamma=0.284;
    int nfast=0;
    for(int i=0;i<2*ndtfast+1;i++){ weight1[i]=0.0; weight2[i]=0.0; }
    scale=(Falpha+1.0)*(Falpha+Fbeta+1.0)/((Falpha+2.0)*(Falpha+Fbeta+2.0)*(double)ndtfast);
    gamma=Fgamma*dmax(0.0,1.0-10.0/(double)ndtfast);
    for(int iter=1;iter<=16;iter++){
        nfast=0;
        for(int i=1;i<=2*ndtfast;i++){
            cff=scale*(double)i;
            weight1[i-1]=(double)(pow(cff,Falpha)-pow(cff,(Falpha+Fbeta)))-gamma*cff;
            if(weight1[i-1]>0.0) nfast=i;
            if((nfast>0)&&(weight1[i-1]<0.0)) weight1[i-1]=0.0;
        }
        wsum=0.0; shift=0.0;
        for(int i=1;i<=nfast;i++){ wsum=wsum+weight1[i-1]; shift=shift+weight1[i-1]*(double)i; }
        scale*=shift/(wsum*(double)ndtfast);
    }
    for(int iter=1;iter<=ndtfast;iter++){
        wsum=0.0; shift=0.0;
        for(int i=1;i<=nfast;i++){ wsum=wsum+weight1[i-1]; shift=shift+(double)i*weight1[i-1]; }
        shift=shift/wsum;
        cff=(double)ndtfast-shift;
        if(cff>1.0){
            nfast=nfast+1;
            for(int i=nfast;i>=2;i--) weight1[i-1]=weight1[i-2];
            weight1[0]=0.0;
        } else if(cff>0.0){
            wsum=1.0-cff;
            for(int i=nfast;i>=2;i--) weight1[i-1]=wsum*weight1[i-1]+cff*weight1[i-2];
            weight1[0]=wsum*weight1[0];
        } else if(cff<-1.0){
            nfast=nfast-1;
            for(int i=1;i<=nfast;i++) weight1[i-1]=weight1[i];
            weight1[nfast]=0.0;
        } else if(cff<0.0){
            wsum=1.0+cff;
            for(int i=1;i<=nfast-1;i++) weight1[i-1]=wsum*weight1[i-1]-cff*weight1[i];
            weight1[nfast-1]=wsum*weight1[nfast-1];
        }
    }
    for(int j=1;j<=nfast;j++){ cff=weight1[j-1]; for(int i=1;i<=j;i++) weight2[i-1]=weight2[i-1]+cff; }
    wsum=0.0; cff=0.0;
    for(int i=1;i<=nfast;i++){ wsum=wsum+weight1[i-1]; cff=cff+weight2[i-1]; }
    wsum=1.0/wsum; cff=1.0/cff;
    for(int i=1;i<=nfast;i++){ weight1[i-1]=wsum*weight1[i-1]; weight2[i-1]=cff*weight2[i-1]; }
    return nfast;
}

/* ------------------------------------------------------------------ stretching curves
 * k-only part of Source/Utils/REMORA_DepthStretchTransform.H:81-139 (N = nz there) */
void o_stretch_coeffs(int k,int N,double theta_s,double theta_b,double *sc_w_o,double *Cs_w_o,double *sc_r_o,double *Cs_r_o)
{
    double ds=1.0/(double)N;
    double Csur,Cbot,sc_r,sc_w,Cs_r,Cs_w;
    if(k==N){ sc_w=0.0; Cs_w=0.0; }
    else if(k==0){ sc_w=-1.0; Cs_w=-1.0; }
    else {
        sc_w=ds*(k-N);
        if(theta_s>0.0) Csur=(1.0-cosh(theta_s*sc_w))/(cosh(theta_s)-1.0);
        else Csur=-sc_w*sc_w;
        if(theta_b>0.0){ Cbot=(exp(theta_b*Csur)-1.0)/(1.0-exp(-theta_b)); Cs_w=Cbot; }
        else Cs_w=Csur;
    }
    sc_r=ds*(k-N+0.5);
    if(theta_s>0.0) Csur=(1.0-cosh(theta_s*sc_r))/(cosh(theta_s)-1.0);
    else Csur=-sc_r*sc_r;
    if(theta_b>0.0){ Cbot=(exp(theta_b*Csur)-1.0)/(1.0-exp(-theta_b)); Cs_r=Cbot; }
    else Cs_r=Csur;
    *sc_w_o=sc_w; *Cs_w_o=Cs_w; *sc_r_o=sc_r; *Cs_r_o=Cs_r;
}
