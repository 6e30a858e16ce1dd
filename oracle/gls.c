/*
 * oracle/gls.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Generic-length-scale turbulence closure, restated pass by pass from
 *   Source/TimeIntegration/REMORA_gls.cpp:5-216   (gls_prestep: predictor of tke and gls on w-points)
 *   Source/TimeIntegration/REMORA_gls.cpp:218-915 (gls_corrector: advection, production/dissipation, implicit vertical
 *                                                  diffusion, stability functions -> Akv, Akt, Akk, Akp, Lscale)
 * Pinned by Tests/REMORA_Gold_Files/Upwelling_GLS (galperin stability, k-epsilon exponents).
 *
 * Index conventions: tke, gls, Akv, Akt, Akk, Akp, Lscale, W, bvf live on w-points k = 0..N+1 (N = nz-1); component 2 of
 * tke/gls holds the predictor, components nstp / nnew the two time levels.
 */
#define ORACLE_SERIAL_LOOPS
#include <stdio.h>
#include "oracle.h"

/* REMORA_gls.cpp:5-210; the ghost fills of :212-215 are done by the caller */
void o_gls_prestep(const Geom *g,Box vbx,Fab gls,Fab tke,Fab W,Fab Huon,Fab Hvom,Fab Hz,Fab pm,Fab pn,Fab msku,Fab mskv,
                   int nstp,int nnew,int iic,int ntfirst,int N,double dt_lev)
{
    Box bx=box_nodal(vbx,2);                         /* tilebox of the w-point MultiFab: k = 0..N+1 */
    Box xbx=box_nodal(bx,0), ybx=box_nodal(bx,1);
    Box xin=box_grow(xbx,0,0,-1), yin=box_grow(ybx,0,0,-1), bin=box_grow(bx,0,0,-1);      /* k = 1..N */
    Box dom=g->domain; int perx=g->per[0], pery=g->per[1];
    Fab XF=fab_alloc(box_growhi(xbx,0,1),1), FX=fab_alloc(box_growhi(xbx,0,1),1), FXL=fab_alloc(box_growhi(xbx,0,1),1);
    Fab EF=fab_alloc(box_growhi(ybx,0,1),1), FE=fab_alloc(box_growhi(ybx,0,1),1), FEL=fab_alloc(box_growhi(ybx,0,1),1);
    Fab Hz_half=fab_alloc(bx,1), CF=fab_alloc(vbx,1), FC=fab_alloc(vbx,1), FCL=fab_alloc(vbx,1);
    fab_setval(XF,0.0); fab_setval(FX,0.0); fab_setval(FXL,0.0); fab_setval(EF,0.0); fab_setval(FE,0.0); fab_setval(FEL,0.0);
    fab_setval(Hz_half,0.0); fab_setval(CF,0.0); fab_setval(FC,0.0); fab_setval(FCL,0.0);                       /* :49-58 */

    FORBOX(xin,i,j,k){                                                                                          /* :74-98 */
        double grad_im1 =(VN(tke,i-1,j,k,nstp)-VN(tke,i-2,j,k,nstp))*V(msku,i-1,j,0);
        double grad_ip1 =(VN(tke,i+1,j,k,nstp)-VN(tke,i  ,j,k,nstp))*V(msku,i+1,j,0);
        double gradL_im1=(VN(gls,i-1,j,k,nstp)-VN(gls,i-2,j,k,nstp))*V(msku,i-1,j,0);
        double gradL_ip1=(VN(gls,i+1,j,k,nstp)-VN(gls,i  ,j,k,nstp))*V(msku,i+1,j,0);
        if(i==dom.lo[0]-1 && !perx){
            grad_im1 =VN(tke,i,j,k,nstp)-VN(tke,i-1,j,k,nstp);
            gradL_im1=VN(gls,i,j,k,nstp)-VN(gls,i-1,j,k,nstp);
        } else if(i==dom.hi[0]+1 && !perx){
            grad_ip1 =VN(tke,i,j,k,nstp)-VN(tke,i-1,j,k,nstp);
            gradL_ip1=VN(gls,i,j,k,nstp)-VN(gls,i-1,j,k,nstp);
        }
        double cff=1.0/6.0;
        V(XF,i,j,k)=0.5*(V(Huon,i,j,k)+V(Huon,i,j,k-1));
        V(FX,i,j,k)=V(XF,i,j,k)*0.5*(VN(tke,i-1,j,k,nstp)+VN(tke,i,j,k,nstp)-cff*(grad_ip1-grad_im1));
        V(FXL,i,j,k)=V(XF,i,j,k)*0.5*(VN(gls,i-1,j,k,nstp)+VN(gls,i,j,k,nstp)-cff*(gradL_ip1-gradL_im1));
    }
    FORBOX(yin,i,j,k){                                                                                          /* :101-125 */
        double grad_jm1 =(VN(tke,i,j-1,k,nstp)-VN(tke,i,j-2,k,nstp))*V(mskv,i,j-1,0);
        double grad_jp1 =(VN(tke,i,j+1,k,nstp)-VN(tke,i,j  ,k,nstp))*V(mskv,i,j+1,0);
        double gradL_jm1=(VN(gls,i,j-1,k,nstp)-VN(gls,i,j-2,k,nstp))*V(mskv,i,j-1,0);
        double gradL_jp1=(VN(gls,i,j+1,k,nstp)-VN(gls,i,j  ,k,nstp))*V(mskv,i,j+1,0);
        if(j==dom.lo[1]-1 && !pery){
            grad_jm1 =VN(tke,i,j,k,nstp)-VN(tke,i,j-1,k,nstp);
            gradL_jm1=VN(gls,i,j,k,nstp)-VN(gls,i,j-1,k,nstp);
        } else if(j==dom.hi[1]+1 && !pery){
            grad_jp1 =VN(tke,i,j,k,nstp)-VN(tke,i,j-1,k,nstp);
            gradL_jp1=VN(gls,i,j,k,nstp)-VN(gls,i,j-1,k,nstp);
        }
        double cff=1.0/6.0;
        V(EF,i,j,k)=0.5*(V(Hvom,i,j,k)+V(Hvom,i,j,k-1));
        V(FE,i,j,k)=V(EF,i,j,k)*0.5*(VN(tke,i,j-1,k,nstp)+VN(tke,i,j,k,nstp)-cff*(grad_jp1-grad_jm1));
        V(FEL,i,j,k)=V(EF,i,j,k)*0.5*(VN(gls,i,j-1,k,nstp)+VN(gls,i,j,k,nstp)-cff*(gradL_jp1-gradL_jm1));
    }
    double gamma=1.0/6.0, cff1,cff2,cff3; int indx;                                                              /* :127-141 */
    if(iic==ntfirst){ cff1=1.0; cff2=0.0; cff3=0.5*dt_lev; indx=nstp; }
    else { cff1=0.5+gamma; cff2=0.5-gamma; cff3=(1.0-gamma)*dt_lev; indx=1-nstp; }
    FORBOX(bin,i,j,k){                                                                                          /* :146-157 */
        double cff=0.5*(V(Hz,i,j,k)+V(Hz,i,j,k-1));
        double cff4=cff3*V(pm,i,j,0)*V(pn,i,j,0);
        V(Hz_half,i,j,k)=cff-cff4*(V(XF,i+1,j,k)-V(XF,i,j,k)+V(EF,i,j+1,k)-V(EF,i,j,k));
        VN(tke,i,j,k,2)=cff*(cff1*VN(tke,i,j,k,nstp)+cff2*VN(tke,i,j,k,indx))-
                        cff4*(V(FX,i+1,j,k)-V(FX,i,j,k)+V(FE,i,j+1,k)-V(FE,i,j,k));
        VN(gls,i,j,k,2)=cff*(cff1*VN(gls,i,j,k,nstp)+cff2*VN(gls,i,j,k,indx))-
                        cff4*(V(FXL,i+1,j,k)-V(FXL,i,j,k)+V(FEL,i,j+1,k)-V(FEL,i,j,k));
        VN(tke,i,j,k,nnew)=cff*VN(tke,i,j,k,nstp);
        VN(gls,i,j,k,nnew)=cff*VN(gls,i,j,k,nstp);
    }
    FORBOX(vbx,i,j,k){                                                                                          /* :161-193 */
        V(CF,i,j,k)=0.5*(V(W,i,j,k+1)+V(W,i,j,k));
        if(k==0){
            double a=1.0/3.0, b=5.0/6.0, c=1.0/6.0;
            V(FC ,i,j,k)=V(CF,i,j,k)*(a*VN(tke,i,j,0,nstp)+b*VN(tke,i,j,1,nstp)-c*VN(tke,i,j,2,nstp));
            V(FCL,i,j,k)=V(CF,i,j,k)*(a*VN(gls,i,j,0,nstp)+b*VN(gls,i,j,1,nstp)-c*VN(gls,i,j,2,nstp));
        } else if(k==N){
            double a=1.0/3.0, b=5.0/6.0, c=1.0/6.0;
            V(FC ,i,j,k)=V(CF,i,j,k)*(a*VN(tke,i,j,k+1,nstp)+b*VN(tke,i,j,k,nstp)-c*VN(tke,i,j,k-1,nstp));
            V(FCL,i,j,k)=V(CF,i,j,k)*(a*VN(gls,i,j,k+1,nstp)+b*VN(gls,i,j,k,nstp)-c*VN(gls,i,j,k-1,nstp));
        } else {
            double a=7.0/12.0, b=1.0/12.0;
            V(FC ,i,j,k)=V(CF,i,j,k)*(a*(VN(tke,i,j,k,nstp)+VN(tke,i,j,k+1,nstp))-b*(VN(tke,i,j,k-1,nstp)+VN(tke,i,j,k+2,nstp)));
            V(FCL,i,j,k)=V(CF,i,j,k)*(a*(VN(gls,i,j,k,nstp)+VN(gls,i,j,k+1,nstp))-b*(VN(gls,i,j,k-1,nstp)+VN(gls,i,j,k+2,nstp)));
        }
    }
    if(iic==ntfirst) cff3=0.5*dt_lev; else cff3=(1.0-gamma)*dt_lev;                                             /* :196-200 */
    FORBOX(bin,i,j,k){                                                                                          /* :202-209 */
        double cff4=cff3*V(pm,i,j,0)*V(pn,i,j,0);
        V(Hz_half,i,j,k)=V(Hz_half,i,j,k)-cff4*(V(CF,i,j,k)-V(CF,i,j,k-1));
        double inv=1.0/V(Hz_half,i,j,k);
        VN(tke,i,j,k,2)=inv*(VN(tke,i,j,k,2)-cff4*(V(FC ,i,j,k)-V(FC ,i,j,k-1)));
        VN(gls,i,j,k,2)=inv*(VN(gls,i,j,k,2)-cff4*(V(FCL,i,j,k)-V(FCL,i,j,k-1)));
    }
    fab_free(&XF); fab_free(&FX); fab_free(&FXL); fab_free(&EF); fab_free(&FE); fab_free(&FEL);
    fab_free(&Hz_half); fab_free(&CF); fab_free(&FC); fab_free(&FCL);
}

/* REMORA_gls.cpp:218-903; the ghost fills of :905-914 are done by the caller */
void o_gls_corrector(const Geom *g,Box vbx,Fab gls,Fab tke,Fab W,Fab Akv,Fab Akt,Fab Akk,Fab Akp,Fab Lscale,Fab mskr,Fab msku,Fab mskv,
                     Fab Hz,Fab u,Fab v,Fab pm,Fab pn,Fab Huon,Fab Hvom,Fab z_w,Fab sustr,Fab svstr,Fab bustr,Fab bvstr,Fab bvf,
                     int nstp,int nnew,int N,double dt_lev)
{
    const GlsParams *G=&g->prm->gls;
    /* constants, :230-349 */
    int Lmy25=(G->p==0.0 && G->n==1.0 && G->m==1.0);
    double vonKar=0.41;                                                           /* REMORA_DataStruct.H:417 */
    double L_sft=vonKar, gls_sigp_cb=G->sigp, ogls_sigp=1.0/gls_sigp_cb;
    double gls_c3m=G->c3m, gls_c3p=G->c3p, gls_cmu0=G->cmu0, gls_m=G->m, gls_n=G->n, gls_p=G->p;
    double gls_Gh0,gls_Ghcri,gls_Ghmin=-0.28,gls_E2=1.33;                         /* REMORA_DataStruct.H:243-268,450-451 */
    double L1=0,L2=0,L3=0,L4=0,L5=0,L6=0,L7=0,L8=0;
    if(G->stab==0){ gls_Gh0=0.0329; gls_Ghcri=0.03;   L1=0.107; L2=0.0032;  L3=0.0864; L4=0.12;  L5=11.9; L6=0.4; L7=0.0; L8=0.48; }
    else if(G->stab==1){ gls_Gh0=0.0444; gls_Ghcri=0.0414; L1=0.127; L2=0.00336; L3=0.0906; L4=0.101; L5=11.2; L6=0.4; L7=0.0; L8=0.318; }
    else { gls_Gh0=0.028; gls_Ghcri=0.02; }
    double Akv_bak=G->Akv_bak, Akt_bak=G->Akt_bak, Akp_bak=G->Akp_bak, Akk_bak=G->Akk_bak;
    double gls_c1=G->c1, gls_c2=G->c2, gls_sigk=G->sigk;
    double sqrt2=sqrt(2.0);
    double cmu_fac1=pow(gls_cmu0,(-gls_p/gls_n));
    double cmu_fac2=pow(gls_cmu0,(3.0+gls_p/gls_n));
    double cmu_fac3=1.0/pow(gls_cmu0,2.0);
    double gls_fac2=pow(gls_cmu0,gls_p)*gls_n*pow(vonKar,gls_n);
    double gls_fac3=pow(gls_cmu0,gls_p)*gls_n;
    double gls_fac4=pow(gls_cmu0,gls_p);
    double gls_fac5=pow(0.56,0.5*gls_n)*pow(gls_cmu0,gls_p);
    double gls_fac6=8.0/pow(gls_cmu0,6.0);
    double gls_exp1=1.0/gls_n, tke_exp1=gls_m/gls_n, tke_exp2=0.5+gls_m/gls_n, tke_exp4=gls_m+0.5*gls_n;
    double gls_s0=0,gls_s1=0,gls_s2=0,gls_s4=0,gls_s5=0,gls_s6=0,gls_b0=0,gls_b1=0,gls_b2=0,gls_b3=0,gls_b4=0,gls_b5=0;
    double my_Sm2=0,my_Sh1=0,my_Sh2=0,my_Sm3=0,my_Sm4=0;
    if(G->stab==0||G->stab==1){                                                   /* Canuto A/B, :290-321 */
        gls_s0=3.0/2.0*L1*L5*L5;
        gls_s1=-L4*(L6+L7)+2.0*L4*L5*(L1-1.0/3.0*L2-L3)+3.0/2.0*L1*L5*L8;
        gls_s2=-3.0/8.0*L1*(L6*L6-L7*L7);
        gls_s4=2.0*L5;
        gls_s5=2.0*L4;
        gls_s6=2.0/3.0*L5*(3.0*L3*L3-L2*L2)-1.0/2.0*L5*L1*(3.0*L3-L2)+3.0/4.0*L1*(L6-L7);
        gls_b0=3.0*L5*L5;
        gls_b1=L5*(7.0*L4+3.0*L8);
        gls_b2=L5*L5*(3.0*L3*L3-L2*L2)-3.0/4.0*(L6*L6-L7*L7);
        gls_b3=L4*(4.0*L4+3.0*L8);
        gls_b5=1.0/4.0*(L2*L2-3.0*L3*L3)*(L6*L6-L7*L7);
        gls_b4=L4*(L2*L6-3.0*L3*L7-L5*(L2*L2-L3*L3))+L5*L8*(3.0*L3*L3-L2*L2);
    } else {                                                                      /* Galperin, :322-343 */
        double A1=0.92,A2=0.74,B1=16.6,B2=10.1,C1=0.08;                           /* REMORA_DataStruct.H:463-467 */
        my_Sm2=9.0*A1*A2;
        my_Sm3=A1*(1.0-3.0*C1-6.0*A1/B1);
        my_Sm4=18.0*A1*A1+9.0*A1*A2;
        my_Sh1=A2*(1.0-6.0*A1/B1);
        my_Sh2=3.0*A2*(6.0*A1+B2);
    }
    double Zos_min=dmax(G->Zos,0.0001), Zob_min=dmax(G->Zob,0.0001), Zos_eff=Zos_min;
    double Gadv=1.0/3.0, eps=1.0e-10;
    double gls_Kmin=G->Kmin, gls_Pmin=G->Pmin;
    Box dom=g->domain; int perx=g->per[0], pery=g->per[1];

    Box bx=box_nodal(vbx,2);                                                      /* w-point tilebox */
    Box gbx1D=box_slab(box_grow(vbx,NGROW-1,NGROW-1,0),2,0), bxD=box_slab(vbx,2,0);
    Box wg=box_grow(bx,NGROW,NGROW,0), cg=box_grow(vbx,NGROW,NGROW,0);
    Fab dU=fab_alloc(wg,1), dV=fab_alloc(wg,1), CF=fab_alloc(wg,1);
    MF mf_shear2_cached=mf_alloc(vbx,0,0,0,1,NGROW,NGROW,0); Fab shear2_cached=mf_shear2_cached.f;
    Fab shear2=fab_alloc(cg,1), buoy2=fab_alloc(cg,1);

    FORBOX2D(gbx1D,i,j){                                                          /* :406-429 */
        V(CF,i,j,0)=0.0; V(dU,i,j,0)=0.0; V(dV,i,j,0)=0.0;
        for(int k=1;k<=N;k++){
            double cff=1.0/(2.0*V(Hz,i,j,k)+V(Hz,i,j,k-1)*(2.0-V(CF,i,j,k-1)));
            V(CF,i,j,k)=cff*V(Hz,i,j,k);
            V(dU,i,j,k)=cff*(3.0*(V(u,i  ,j,k)-V(u,i  ,j,k-1)+V(u,i+1,j,k)-V(u,i+1,j,k-1))-V(Hz,i,j,k-1)*V(dU,i,j,k-1));
            V(dV,i,j,k)=cff*(3.0*(V(v,i,j  ,k)-V(v,i,j  ,k-1)+V(v,i,j+1,k)-V(v,i,j+1,k-1))-V(Hz,i,j,k-1)*V(dV,i,j,k-1));
        }
        V(dU,i,j,N+1)=0.0; V(dV,i,j,N+1)=0.0;
        for(int k=N;k>=1;k--){
            V(dU,i,j,k)=V(dU,i,j,k)-V(CF,i,j,k)*V(dU,i,j,k+1);
            V(dV,i,j,k)=V(dV,i,j,k)-V(CF,i,j,k)*V(dV,i,j,k+1);
        }
        V(shear2_cached,i,j,0)=0.0;
        for(int k=1;k<=N;k++) V(shear2_cached,i,j,k)=V(dU,i,j,k)*V(dU,i,j,k)+V(dV,i,j,k)*V(dV,i,j,k);
    }
    o_physbc(g,&mf_shear2_cached,mskr,0,1,BCV_FOEXTRAP,0);                        /* :434: foextrap on every side, periodic ones too */
    fab_setval(CF,0.0);                                                           /* :435 */

    Box xbx=box_nodal(bx,0), ybx=box_nodal(bx,1), gbx1=box_grow(bx,NGROW-1,NGROW-1,0);
    Box xin=box_grow(xbx,0,0,-1), yin=box_grow(ybx,0,0,-1), bin=box_grow(bx,0,0,-1);
    Box bx_growloxy=box_growlo(box_growlo(bin,0,1),1,1);
    Fab tmp_buoy=fab_alloc(bx_growloxy,1), tmp_shear=fab_alloc(bx_growloxy,1);
    Fab curvK=fab_alloc(gbx1,1), curvP=fab_alloc(gbx1,1);
    Fab FXK=fab_alloc(xin,1), FXP=fab_alloc(xin,1), FEK=fab_alloc(yin,1), FEP=fab_alloc(yin,1);
    Fab FCK=fab_alloc(vbx,1), FCP=fab_alloc(vbx,1), BCK=fab_alloc(vbx,1), BCP=fab_alloc(vbx,1);
    fab_setval(tmp_buoy,0.0); fab_setval(tmp_shear,0.0); fab_setval(curvK,0.0); fab_setval(curvP,0.0);
    fab_setval(FXK,0.0); fab_setval(FXP,0.0); fab_setval(FEK,0.0); fab_setval(FEP,0.0);
    fab_setval(FCK,0.0); fab_setval(FCP,0.0); fab_setval(BCK,0.0); fab_setval(BCP,0.0);                         /* :487-501 */

    FORBOX(bx_growloxy,i,j,k){                                                    /* :527-531 */
        V(tmp_buoy ,i,j,k)=0.25*(V(bvf,i,j,k)+V(bvf,i+1,j,k)+V(bvf,i,j+1,k)+V(bvf,i+1,j+1,k));
        V(tmp_shear,i,j,k)=0.25*(V(shear2_cached,i,j,k)+V(shear2_cached,i+1,j,k)+V(shear2_cached,i,j+1,k)+V(shear2_cached,i+1,j+1,k));
    }
    FORBOX(bin,i,j,k){                                                            /* :533-537 */
        V(buoy2 ,i,j,k)=0.25*(V(tmp_buoy ,i,j,k)+V(tmp_buoy ,i-1,j,k)+V(tmp_buoy ,i,j-1,k)+V(tmp_buoy ,i-1,j-1,k));
        V(shear2,i,j,k)=0.25*(V(tmp_shear,i,j,k)+V(tmp_shear,i-1,j,k)+V(tmp_shear,i,j-1,k)+V(tmp_shear,i-1,j-1,k));
    }
    { Box b=box_growlo(xin,0,1);                                                  /* :540-563 */
      FORBOX(b,i,j,k){
        double gradK,gradK_ip1,gradP,gradP_ip1;
        if(i==dom.lo[0]-1 && !perx){
            gradK_ip1=VN(tke,i+1,j,k,2)-VN(tke,i,j,k,2); gradK=gradK_ip1;
            gradP_ip1=VN(gls,i+1,j,k,2)-VN(gls,i,j,k,2); gradP=gradP_ip1;
        } else if(i==dom.hi[0]+1 && !perx){
            gradK=VN(tke,i,j,k,2)-VN(tke,i-1,j,k,2); gradK_ip1=gradK;
            gradP=VN(gls,i,j,k,2)-VN(gls,i-1,j,k,2); gradP_ip1=gradP;
        } else {
            gradK    =(VN(tke,i  ,j,k,2)-VN(tke,i-1,j,k,2))*V(msku,i  ,j,0);
            gradK_ip1=(VN(tke,i+1,j,k,2)-VN(tke,i  ,j,k,2))*V(msku,i+1,j,0);
            gradP    =(VN(gls,i  ,j,k,2)-VN(gls,i-1,j,k,2))*V(msku,i  ,j,0);
            gradP_ip1=(VN(gls,i+1,j,k,2)-VN(gls,i  ,j,k,2))*V(msku,i+1,j,0);
        }
        V(curvK,i,j,k)=gradK_ip1-gradK;
        V(curvP,i,j,k)=gradP_ip1-gradP;
      } }
    FORBOX(xin,i,j,k){                                                            /* :564-572 */
        double cff=0.5*(V(Huon,i,j,k)+V(Huon,i,j,k-1));
        double c1=(cff>0.0)?V(curvK,i-1,j,k):V(curvK,i,j,k);
        double c2=(cff>0.0)?V(curvP,i-1,j,k):V(curvP,i,j,k);
        V(FXK,i,j,k)=cff*0.5*(VN(tke,i-1,j,k,2)+VN(tke,i,j,k,2)-Gadv*c1);
        V(FXP,i,j,k)=cff*0.5*(VN(gls,i-1,j,k,2)+VN(gls,i,j,k,2)-Gadv*c2);
    }
    { Box b=box_growlo(yin,1,1);                                                  /* :575-593 */
      FORBOX(b,i,j,k){
        double gradK    =(VN(tke,i,j  ,k,2)-VN(tke,i,j-1,k,2))*V(mskv,i,j  ,0);
        double gradK_jp1=(VN(tke,i,j+1,k,2)-VN(tke,i,j  ,k,2))*V(mskv,i,j+1,0);
        double gradP    =(VN(gls,i,j  ,k,2)-VN(gls,i,j-1,k,2))*V(mskv,i,j  ,0);
        double gradP_jp1=(VN(gls,i,j+1,k,2)-VN(gls,i,j  ,k,2))*V(mskv,i,j+1,0);
        if(j==dom.lo[1]-1 && !pery){ gradK=gradK_jp1; gradP=gradP_jp1; }
        else if(j==dom.hi[1]+1 && !pery){ gradK_jp1=gradK; gradP_jp1=gradP; }
        V(curvK,i,j,k)=gradK_jp1-gradK;
        V(curvP,i,j,k)=gradP_jp1-gradP;
      } }
    FORBOX(yin,i,j,k){                                                            /* :594-602 */
        double cff=0.5*(V(Hvom,i,j,k)+V(Hvom,i,j,k-1));
        double c1=(cff>0.0)?V(curvK,i,j-1,k):V(curvK,i,j,k);
        double c2=(cff>0.0)?V(curvP,i,j-1,k):V(curvP,i,j,k);
        V(FEK,i,j,k)=cff*0.5*(VN(tke,i,j-1,k,2)+VN(tke,i,j,k,2)-Gadv*c1);
        V(FEP,i,j,k)=cff*0.5*(VN(gls,i,j-1,k,2)+VN(gls,i,j,k,2)-Gadv*c2);
    }
    FORBOX(bin,i,j,k){                                                            /* :606-616 */
        double cff=dt_lev*V(pm,i,j,0)*V(pn,i,j,0);
        VN(tke,i,j,k,nnew)=VN(tke,i,j,k,nnew)-cff*(V(FXK,i+1,j,k)-V(FXK,i,j,k)+V(FEK,i,j+1,k)-V(FEK,i,j,k));
        VN(tke,i,j,k,nnew)=dmax(VN(tke,i,j,k,nnew),gls_Kmin);
        VN(gls,i,j,k,nnew)=VN(gls,i,j,k,nnew)-cff*(V(FXP,i+1,j,k)-V(FXP,i,j,k)+V(FEP,i,j+1,k)-V(FEP,i,j,k));
        VN(gls,i,j,k,nnew)=dmax(VN(gls,i,j,k,nnew),gls_Pmin);
    }
    FORBOX2D(bxD,i,j){                                                            /* :620-640 vertical advection fluxes */
        double a=7.0/12.0, b=1.0/12.0;
        for(int k=1;k<=N-1;k++){
            double cff=0.5*(V(W,i,j,k+1)+V(W,i,j,k));
            V(FCK,i,j,k)=cff*(a*(VN(tke,i,j,k,2)+VN(tke,i,j,k+1,2))-b*(VN(tke,i,j,k-1,2)+VN(tke,i,j,k+2,2)));
            V(FCP,i,j,k)=cff*(a*(VN(gls,i,j,k,2)+VN(gls,i,j,k+1,2))-b*(VN(gls,i,j,k-1,2)+VN(gls,i,j,k+2,2)));
        }
        a=1.0/3.0; b=5.0/6.0; double c=1.0/6.0;
        double cff=0.5*(V(W,i,j,0)+V(W,i,j,1));
        V(FCK,i,j,0)=cff*(a*VN(tke,i,j,0,2)+b*VN(tke,i,j,1,2)-c*VN(tke,i,j,2,2));
        V(FCP,i,j,0)=cff*(a*VN(gls,i,j,0,2)+b*VN(gls,i,j,1,2)-c*VN(gls,i,j,2,2));
        cff=0.5*(V(W,i,j,N+1)+V(W,i,j,N));
        V(FCK,i,j,N)=cff*(a*VN(tke,i,j,N+1,2)+b*VN(tke,i,j,N,2)-c*VN(tke,i,j,N-1,2));
        V(FCP,i,j,N)=cff*(a*VN(gls,i,j,N+1,2)+b*VN(gls,i,j,N,2)-c*VN(gls,i,j,N-1,2));
    }
    FORBOX(bin,i,j,k){                                                            /* :641-648 */
        double cff=dt_lev*V(pm,i,j,0)*V(pn,i,j,0);
        VN(tke,i,j,k,nnew)=VN(tke,i,j,k,nnew)-cff*(V(FCK,i,j,k)-V(FCK,i,j,k-1));
        VN(tke,i,j,k,nnew)=dmax(VN(tke,i,j,k,nnew),gls_Kmin);
        VN(gls,i,j,k,nnew)=VN(gls,i,j,k,nnew)-cff*(V(FCP,i,j,k)-V(FCP,i,j,k-1));
        VN(gls,i,j,k,nnew)=dmax(VN(gls,i,j,k,nnew),gls_Pmin);
    }
    { double cff=-0.5*dt_lev;                                                     /* :653-663 */
      FORBOX(vbx,i,j,k){
        if(k==0||k==N){ V(FCK,i,j,k)=0.0; V(FCP,i,j,k)=0.0; }
        else { V(FCK,i,j,k)=cff*(V(Akk,i,j,k)+V(Akk,i,j,k+1))/V(Hz,i,j,k);
               V(FCP,i,j,k)=cff*(V(Akp,i,j,k)+V(Akp,i,j,k+1))/V(Hz,i,j,k); }
      } }
    FORBOX(bin,i,j,k){                                                            /* :665-719 production / dissipation */
        double strat2=V(buoy2,i,j,k);
        double gls_c3=(strat2>0.0)?gls_c3m:gls_c3p;
        double Kprod=V(shear2,i,j,k)*(V(Akv,i,j,k)-Akv_bak)-strat2*(VN(Akt,i,j,k,0)-Akt_bak);
        double Pprod=gls_c1*V(shear2,i,j,k)*(V(Akv,i,j,k)-Akv_bak)-gls_c3*strat2*(VN(Akt,i,j,k,0)-Akt_bak);
        double on1=(Kprod<0.0)?0.0:1.0, on2=(Pprod<0.0)?0.0:1.0;
        if(Kprod<0.0) Kprod=Kprod+strat2*(VN(Akt,i,j,k,0)-Akt_bak);
        if(Pprod<0.0) Pprod=Pprod+gls_c3*strat2*(VN(Akt,i,j,k,0)-Akt_bak);
        double cff_Hz=0.5*(V(Hz,i,j,k)+V(Hz,i,j,k-1));
        VN(tke,i,j,k,nnew)=VN(tke,i,j,k,nnew)+dt_lev*cff_Hz*Kprod;
        VN(gls,i,j,k,nnew)=VN(gls,i,j,k,nnew)+dt_lev*cff_Hz*Pprod*VN(gls,i,j,k,nstp)/dmax(VN(tke,i,j,k,nstp),gls_Kmin);
        double wall_fac=1.0;
        if(Lmy25){
            wall_fac=1.0+gls_E2/(vonKar*vonKar)*
                     pow(pow(VN(gls,i,j,k,nstp),(gls_exp1))*cmu_fac1*pow(VN(tke,i,j,k,nstp),-tke_exp1)*
                         (1.0/(V(z_w,i,j,k)-V(z_w,i,j,0))),2.0)+
                     0.25/(vonKar*vonKar)*
                     pow(pow(VN(gls,i,j,k,nstp),gls_exp1)*cmu_fac1*pow(VN(tke,i,j,k,nstp),-tke_exp1)*
                         (1.0/(V(z_w,i,j,N+1)-V(z_w,i,j,k))),2.0);
        }
        V(BCK,i,j,k)=cff_Hz*(1.0+dt_lev*pow(VN(gls,i,j,k,nstp),(-gls_exp1))*cmu_fac2*pow(VN(tke,i,j,k,nstp),tke_exp2)+
                             dt_lev*(1.0-on1)*strat2*(VN(Akt,i,j,k,0)-Akt_bak)/VN(tke,i,j,k,nstp))-
                     V(FCK,i,j,k)-V(FCK,i,j,k-1);
        V(BCP,i,j,k)=cff_Hz*(1.0+dt_lev*gls_c2*wall_fac*pow(VN(gls,i,j,k,nstp),-gls_exp1)*cmu_fac2*pow(VN(tke,i,j,k,nstp),tke_exp2)+
                             dt_lev*(1.0-on2)*gls_c3*strat2*(VN(Akt,i,j,k,0)-Akt_bak)/VN(tke,i,j,k,nstp))-
                     V(FCP,i,j,k)-V(FCP,i,j,k-1);
    }
    FORBOX2D(bxD,i,j){                                                            /* :721-788 boundary values and the two Thomas solves */
        VN(tke,i,j,N+1,nnew)=dmax(cmu_fac3*0.5*sqrt((V(sustr,i,j,0)+V(sustr,i+1,j,0))*(V(sustr,i,j,0)+V(sustr,i+1,j,0))+
                                                   (V(svstr,i,j,0)+V(svstr,i,j+1,0))*(V(svstr,i,j,0)+V(svstr,i,j+1,0))),gls_Kmin);
        VN(tke,i,j,0,nnew)  =dmax(cmu_fac3*0.5*sqrt((V(bustr,i,j,0)+V(bustr,i+1,j,0))*(V(bustr,i,j,0)+V(bustr,i+1,j,0))+
                                                   (V(bvstr,i,j,0)+V(bvstr,i,j+1,0))*(V(bvstr,i,j,0)+V(bvstr,i,j+1,0))),gls_Kmin);
        VN(gls,i,j,N+1,nnew)=dmax(pow(gls_cmu0,gls_p)*pow(VN(tke,i,j,N+1,nnew),gls_m)*pow(L_sft*Zos_eff,gls_n),gls_Pmin);
        double cff_gls=gls_fac4*pow(vonKar*Zob_min,gls_n);
        VN(gls,i,j,0,nnew)=dmax(cff_gls*pow(VN(tke,i,j,0,nnew),(gls_m)),gls_Pmin);

        double tke_fluxt=0.0, tke_fluxb=0.0;
        double cb=1.0/V(BCK,i,j,N);
        V(CF,i,j,N)=cb*V(FCK,i,j,N-1);
        VN(tke,i,j,N,nnew)=cb*(VN(tke,i,j,N,nnew)+tke_fluxt);
        for(int k=N-1;k>=1;k--){
            cb=1.0/(V(BCK,i,j,k)-V(CF,i,j,k+1)*V(FCK,i,j,k));
            V(CF,i,j,k)=cb*V(FCK,i,j,k-1);
            VN(tke,i,j,k,nnew)=cb*(VN(tke,i,j,k,nnew)-V(FCK,i,j,k)*VN(tke,i,j,k+1,nnew));
        }
        VN(tke,i,j,1,nnew)=VN(tke,i,j,1,nnew)-cb*tke_fluxb;
        VN(tke,i,j,1,nnew)=dmax(VN(tke,i,j,1,nnew),gls_Kmin);
        for(int k=2;k<=N;k++){
            VN(tke,i,j,k,nnew)=VN(tke,i,j,k,nnew)-V(CF,i,j,k)*VN(tke,i,j,k-1,nnew);
            VN(tke,i,j,k,nnew)=dmax(VN(tke,i,j,k,nnew),gls_Kmin);
        }
        double ct=0.5*(VN(tke,i,j,N+1,nnew)+VN(tke,i,j,N,nnew));
        double gls_fluxt=dt_lev*gls_fac3*pow(ct,gls_m)*pow(L_sft,(gls_n))*pow(Zos_eff+0.5*V(Hz,i,j,N),gls_n-1.0)*
                         0.5*(V(Akp,i,j,N+1)+V(Akp,i,j,N));
        ct=0.5*(VN(tke,i,j,0,nnew)+VN(tke,i,j,1,nnew));
        double gls_fluxb=dt_lev*gls_fac2*pow(ct,gls_m)*pow(0.5*V(Hz,i,j,0)+Zob_min,gls_n-1.0)*0.5*(V(Akp,i,j,0)+V(Akp,i,j,1));
        double cp=1.0/V(BCP,i,j,N);
        V(CF,i,j,N)=cp*V(FCP,i,j,N-1);
        VN(gls,i,j,N,nnew)=cp*(VN(gls,i,j,N,nnew)-gls_fluxt);
        for(int k=N-1;k>=1;k--){
            cp=1.0/(V(BCP,i,j,k)-V(CF,i,j,k+1)*V(FCP,i,j,k));
            V(CF,i,j,k)=cp*V(FCP,i,j,k-1);
            VN(gls,i,j,k,nnew)=cp*(VN(gls,i,j,k,nnew)-V(FCP,i,j,k)*VN(gls,i,j,k+1,nnew));
        }
        VN(gls,i,j,1,nnew)=VN(gls,i,j,1,nnew)-cp*gls_fluxb;
        for(int k=2;k<=N;k++) VN(gls,i,j,k,nnew)=VN(gls,i,j,k,nnew)-V(CF,i,j,k)*VN(gls,i,j,k-1,nnew);
    }
    FORBOX(bin,i,j,k){                                                            /* :791-885 limits, stability functions, mixing */
        VN(tke,i,j,k,nnew)=dmax(VN(tke,i,j,k,nnew),gls_Kmin);
        VN(gls,i,j,k,nnew)=dmax(VN(gls,i,j,k,nnew),gls_Pmin);
        if(gls_n>=0.0)
            VN(gls,i,j,k,nnew)=dmin(VN(gls,i,j,k,nnew),gls_fac5*pow(VN(tke,i,j,k,nnew),tke_exp4)*
                                    pow(sqrt(dmax(0.0,V(buoy2,i,j,k)))+eps,-gls_n));
        else
            VN(gls,i,j,k,nnew)=dmax(VN(gls,i,j,k,nnew),gls_fac5*pow(VN(tke,i,j,k,nnew),(tke_exp4))*
                                    pow((sqrt(dmax(0.0,V(buoy2,i,j,k)))+eps),(-gls_n)));
        double Ls_lmt;
        double Ls_unlmt=dmax(eps,pow(VN(gls,i,j,k,nnew),(gls_exp1))*cmu_fac1*pow(VN(tke,i,j,k,nnew),(-tke_exp1)));
        if(V(buoy2,i,j,k)>0.0) Ls_lmt=dmin(Ls_unlmt,sqrt(0.56*VN(tke,i,j,k,nnew)/(dmax(0.0,V(buoy2,i,j,k))+eps)));
        else Ls_lmt=Ls_unlmt;
        VN(gls,i,j,k,nnew)=dmax(pow(gls_cmu0,gls_p)*pow(VN(tke,i,j,k,nnew),gls_m)*pow(Ls_lmt,gls_n),gls_Pmin);
        double Sh=0.0,Sm=0.0;
        double Gh=dmin(gls_Gh0,-V(buoy2,i,j,k)*Ls_lmt*Ls_lmt/(2.0*VN(tke,i,j,k,nnew)));
        Gh=dmin(Gh,Gh-(Gh-gls_Ghcri)*(Gh-gls_Ghcri)/(Gh+gls_Gh0-2.0*gls_Ghcri));
        Gh=dmax(Gh,gls_Ghmin);
        if(G->stab==0||G->stab==1){
            double Gm=(gls_b0/gls_fac6-gls_b1*Gh+gls_b3*gls_fac6*(Gh*Gh))/(gls_b2-gls_b4*gls_fac6*Gh);
            Gm=dmin(Gm,V(shear2,i,j,k)*Ls_lmt*Ls_lmt/(2.0*VN(tke,i,j,k,nnew)));
            double sc=gls_b0-gls_b1*gls_fac6*Gh+gls_b2*gls_fac6*Gm+gls_b3*gls_fac6*gls_fac6*Gh*Gh-
                      gls_b4*gls_fac6*gls_fac6*Gh*Gm+gls_b5*gls_fac6*gls_fac6*Gm*Gm;
            Sm=(gls_s0-gls_s1*gls_fac6*Gh+gls_s2*gls_fac6*Gm)/sc;
            Sh=(gls_s4-gls_s5*gls_fac6*Gh+gls_s6*gls_fac6*Gm)/sc;
            Sm=dmax(Sm,0.0); Sh=dmax(Sh,0.0);
            double c3=gls_cmu0*gls_cmu0*gls_cmu0;
            Sm=Sm*sqrt2/(c3); Sh=Sh*sqrt2/c3;
        } else {
            double cg_=1.0-my_Sh2*Gh;
            Sh=my_Sh1/cg_;
            Sm=(my_Sm3+Sh*Gh*my_Sm4)/(1.0-my_Sm2*Gh);
        }
        double ql=sqrt2*0.5*(Ls_lmt*sqrt(VN(tke,i,j,k,nnew))+V(Lscale,i,j,k)*sqrt(VN(tke,i,j,k,nstp)));
        V(Akv,i,j,k)=Akv_bak+Sm*ql;
        for(int n=0;n<NCONS;n++) VN(Akt,i,j,k,n)=Akt_bak+Sh*ql;
        V(Akk,i,j,k)=Akk_bak+Sm*ql/gls_sigk;
        V(Akp,i,j,k)=Akp_bak+Sm*ql*ogls_sigp;
        V(Lscale,i,j,k)=Ls_lmt;
    }
    FORBOX2D(bxD,i,j){                                                            /* :886-902 */
        V(Akv,i,j,N+1)=Akv_bak+L_sft*Zos_eff*gls_cmu0*sqrt(VN(tke,i,j,N+1,nnew));
        V(Akv,i,j,0)  =Akv_bak+vonKar*Zob_min*gls_cmu0*sqrt(VN(tke,i,j,0,nnew));
        V(Akk,i,j,N+1)=Akk_bak+V(Akv,i,j,N+1)/gls_sigk;
        V(Akk,i,j,0)  =Akk_bak+V(Akv,i,j,0)/gls_sigk;
        V(Akp,i,j,N+1)=Akp_bak+V(Akv,i,j,N+1)*ogls_sigp;
        V(Akp,i,j,0)  =Akp_bak+V(Akv,i,j,0)/gls_sigp_cb;
        for(int n=0;n<NCONS;n++){ VN(Akt,i,j,N+1,n)=Akt_bak; VN(Akt,i,j,0,n)=Akt_bak; }
    }
    (void)gls_E2; (void)z_w;
    fab_free(&dU); fab_free(&dV); fab_free(&CF); mf_free(&mf_shear2_cached); fab_free(&shear2); fab_free(&buoy2);
    fab_free(&tmp_buoy); fab_free(&tmp_shear); fab_free(&curvK); fab_free(&curvP);
    fab_free(&FXK); fab_free(&FXP); fab_free(&FEK); fab_free(&FEP); fab_free(&FCK); fab_free(&FCP); fab_free(&BCK); fab_free(&BCP);
}
