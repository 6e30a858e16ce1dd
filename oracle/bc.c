/*
 * oracle/bc.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Ghost-cell fills for a single box covering the whole domain:
 *   - periodic FillBoundary (what amrex::FillPatchSingleLevel reduces to when source == destination,
 *     Source/BoundaryConditions/REMORA_FillPatch.cpp:96-102)
 *   - REMORAPhysBCFunct for the BC kinds the four pinned cases use (periodic + SlipWall => int_dir, foextrap, ext_dir):
 *     Source/BoundaryConditions/REMORA_PhysBCFunct.cpp:21-93, REMORA_BoundaryConditions_{cons,xvel,yvel,zvel}.cpp
 *   - BC table: Source/Initialization/REMORA_init_bcs.cpp:9-679
 */
#define ORACLE_SERIAL_LOOPS
#include <stdio.h>
#include "oracle.h"

MF mf_alloc(Box cb,int ix0,int ix1,int ix2,int nc,int ng0,int ng1,int ng2)
{
    MF m; m.ix[0]=ix0; m.ix[1]=ix1; m.ix[2]=ix2; m.ng[0]=ng0; m.ng[1]=ng1; m.ng[2]=ng2;
    m.valid=cb; for(int d=0;d<3;d++) if(m.ix[d]) m.valid.hi[d]+=1;
    m.f=fab_alloc(box_grow(m.valid,ng0,ng1,ng2),nc);
    return m;
}
void mf_free(MF *m){ fab_free(&m->f); }

static void bc_set(BCRec *b,int dir,int hi,int v){ if(hi) b->hi[dir]=v; else b->lo[dir]=v; }
static void bc_abort(const char *m){ fprintf(stderr,"oracle: %s\n",m); abort(); }

/* REMORA::init_bcs, Source/Initialization/REMORA_init_bcs.cpp:9-679 */
void o_init_bcs(Geom *g)
{
    const Params *P=g->prm;
    for(int v=0;v<BCV_NUM;v++) for(int d=0;d<3;d++){ g->bcs[v].lo[d]=BC_BOGUS; g->bcs[v].hi[d]=BC_BOGUS; }
    for(int v=0;v<BCV_NUM;v++) for(int o=0;o<6;o++) g->extdir[v][o]=0.0;
    for(int o=0;o<6;o++){ g->extdir[BCV_TEMP][o]=1.e19; g->extdir[BCV_SALT][o]=1.e20; g->extdir[BCV_SCALAR][o]=1.e21; }   /* :161-163 */
    /* phys_bc_type (:9-123, :177-199): one kind per variable and face; z faces default to SlipWall (:130) */
    for(int v=0;v<BCV_USER;v++) for(int o=0;o<6;o++){
        int dir=o%3, kind;
        if(dir==2) kind=PBC_SLIP_WALL;
        else if(P->bc_set) kind=P->bc_kind[v][o];
        else kind= P->periodic[dir] ? PBC_PERIODIC : PBC_SLIP_WALL;
        if(dir<2 && P->periodic[dir]){                                                        /* :107-115 */
            if(kind==PBC_UNDEFINED) kind=PBC_PERIODIC;
            else if(kind!=PBC_PERIODIC) bc_abort("Wrong BC type for periodic boundary");
        } else if(kind==PBC_PERIODIC) bc_abort("Periodic boundary specified in a non-periodic direction");   /* :96-98 */
        if(kind<0||kind>=PBC_UNDEFINED) bc_abort("This BC type is unknown");                    /* :117-121 */
        if(kind==PBC_CHAPMAN && v!=BCV_ZETA) bc_abort("Chapman BC can only be applied to zeta");                     /* :71-73 */
        if(kind==PBC_FLATHER && !(v==BCV_UBAR||v==BCV_VBAR)) bc_abort("Flather BC can only be applied to ubar or vbar"); /* :80-82 */
        g->phys[v][o]=kind;
        if(P->bc_set && dir<2){
            int isvel=(v==BCV_XVEL||v==BCV_YVEL||v==BCV_ZVEL);
            if(kind==PBC_INFLOW && (isvel||v==BCV_SCALAR)) g->extdir[v][o]=P->bc_val[v][o];     /* :27-36 */
            if(kind==PBC_NO_SLIP_WALL && isvel) g->extdir[v][o]= (v-BCV_XVEL==dir) ? 0.0 : P->bc_val[v][o];   /* :43-54 */
        }
    }
    for(int o=0;o<6;o++){                                                                     /* :210-318 velocities */
        int dir=o%3, hi=o/3;
        for(int i=0;i<2;i++){
            int bct=g->phys[BCV_XVEL+i][o];
            switch(bct){
            case PBC_SYMMETRY:     bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_REFLECT_EVEN); if(i==1) bc_set(&g->bcs[BCV_XVEL+dir],dir,hi,BC_REFLECT_ODD); break;
            case PBC_OUTFLOW:      bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_FOEXTRAP); break;
            case PBC_INFLOW:       bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_EXT_DIR); break;
            case PBC_NO_SLIP_WALL: bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_EXT_DIR); break;
            case PBC_SLIP_WALL:    bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_FOEXTRAP); if(i==1) bc_set(&g->bcs[BCV_XVEL+dir],dir,hi,BC_EXT_DIR); break;
            case PBC_PERIODIC:     bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_INT_DIR); break;
            case PBC_CLAMPED:      bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_CLAMPED); break;
            case PBC_ORLANSKI_RAD: bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_ORLANSKI_RAD); break;
            case PBC_ORLANSKI_RAD_NUDGE: bc_set(&g->bcs[BCV_XVEL+i],dir,hi,BC_ORLANSKI_RAD_NUDGE); break;
            default: bc_abort("Velocity boundary condition not validly specified");
            }
        }
        bc_set(&g->bcs[BCV_ZVEL],dir,hi,BC_FOEXTRAP);                                         /* :311-316 */
    }
    for(int o=0;o<6;o++){                                                                     /* :326-411 tracers */
        int dir=o%3, hi=o/3;
        for(int i=0;i<NCONS;i++){
            int t;
            switch(g->phys[BCV_CONS+i][o]){
            case PBC_SYMMETRY: t=BC_REFLECT_EVEN; break;
            case PBC_OUTFLOW: case PBC_NO_SLIP_WALL: case PBC_SLIP_WALL: t=BC_FOEXTRAP; break;
            case PBC_INFLOW: t=BC_EXT_DIR; break;
            case PBC_PERIODIC: t=BC_INT_DIR; break;
            case PBC_CLAMPED: t=BC_CLAMPED; break;
            case PBC_ORLANSKI_RAD: t=BC_ORLANSKI_RAD; break;
            case PBC_ORLANSKI_RAD_NUDGE: t=BC_ORLANSKI_RAD_NUDGE; break;
            default: t=0; bc_abort("Scalar/tracer boundary condition not validly specified");
            }
            bc_set(&g->bcs[BCV_CONS+i],dir,hi,t);
        }
    }
    for(int o=0;o<6;o++){                                                                     /* :420-543 ubar, vbar and the simplified 2-D pair */
        int dir=o%3, hi=o/3;
        for(int i=0;i<2;i++){
            BCRec *b=&g->bcs[BCV_UBAR+i], *b2=&g->bcs[BCV_U2D_SIMPLE+i];
            switch(g->phys[BCV_UBAR+i][o]){
            case PBC_SYMMETRY:
                bc_set(b,dir,hi,BC_REFLECT_EVEN); bc_set(b2,dir,hi,BC_REFLECT_EVEN);
                if(i==1&&dir!=2){ bc_set(&g->bcs[BCV_UBAR+dir],dir,hi,BC_REFLECT_ODD); bc_set(&g->bcs[BCV_U2D_SIMPLE+dir],dir,hi,BC_REFLECT_ODD); }
                break;
            case PBC_OUTFLOW:      bc_set(b,dir,hi,BC_FOEXTRAP); bc_set(b2,dir,hi,BC_FOEXTRAP); break;
            case PBC_INFLOW:       bc_set(b,dir,hi,BC_EXT_DIR);  bc_set(b2,dir,hi,BC_EXT_DIR); break;
            case PBC_NO_SLIP_WALL: bc_set(b,dir,hi,BC_EXT_DIR);  bc_set(b2,dir,hi,BC_EXT_DIR); break;
            case PBC_SLIP_WALL:
                bc_set(b,dir,hi,BC_FOEXTRAP); bc_set(b2,dir,hi,BC_FOEXTRAP);
                if(i==1&&dir!=2){ bc_set(&g->bcs[BCV_UBAR+dir],dir,hi,BC_EXT_DIR); bc_set(&g->bcs[BCV_U2D_SIMPLE+dir],dir,hi,BC_EXT_DIR); }
                break;
            case PBC_PERIODIC:     bc_set(b,dir,hi,BC_INT_DIR);  bc_set(b2,dir,hi,BC_INT_DIR); break;
            case PBC_CLAMPED:      bc_set(b,dir,hi,BC_CLAMPED);  bc_set(b2,dir,hi,BC_FOEXTRAP); break;
            case PBC_FLATHER:
                bc_set(b,dir,hi,BC_CHAPMAN); bc_set(b2,dir,hi,BC_FOEXTRAP);
                if(i==1&&dir!=2){ bc_set(&g->bcs[BCV_UBAR+dir],dir,hi,BC_FLATHER); bc_set(&g->bcs[BCV_U2D_SIMPLE+dir],dir,hi,BC_FOEXTRAP); }
                break;
            default: bc_abort("ubar or vbar boundary condition not validly specified");
            }
        }
    }
    for(int o=0;o<6;o++){                                                                     /* :551-628 zeta, tke */
        int dir=o%3, hi=o/3;
        for(int i=0;i<2;i++){
            int t;
            switch(g->phys[BCV_ZETA+i][o]){
            case PBC_SYMMETRY: t=BC_REFLECT_EVEN; break;
            case PBC_OUTFLOW: case PBC_NO_SLIP_WALL: case PBC_SLIP_WALL: t=BC_FOEXTRAP; break;
            case PBC_INFLOW: t=BC_EXT_DIR; break;
            case PBC_PERIODIC: t=BC_INT_DIR; break;
            case PBC_CHAPMAN: t=BC_CHAPMAN; break;
            case PBC_CLAMPED: t=BC_CLAMPED; break;
            default: t=0; bc_abort("Free surface (zeta) boundary condition not validly specified");
            }
            bc_set(&g->bcs[BCV_ZETA+i],dir,hi,t);
        }
    }
    for(int o=0;o<6;o++){                                                                     /* :636-667 */
        bc_set(&g->bcs[BCV_FOEXTRAP_PERIODIC],o%3,o/3,BC_FOEXTRAP);
        bc_set(&g->bcs[BCV_FOEXTRAP],o%3,o/3,BC_FOEXTRAP);
    }
}
void o_get_bc(const Geom *g,int bccomp,int *out6){ for(int d=0;d<3;d++){ out6[d]=g->bcs[bccomp].lo[d]; out6[3+d]=g->bcs[bccomp].hi[d]; } }

/* periodic fill of ghost points from valid points of the same (single) box */
void o_fill_boundary(const Geom *g,MF *mf,int icomp,int ncomp)
{
    Fab f=mf->f; Box gb=fab_box(f), vb=mf->valid;
    int len[2]={ g->domain.hi[0]-g->domain.lo[0]+1, g->domain.hi[1]-g->domain.lo[1]+1 };
    if(!g->per[0] && !g->per[1]) return;
    for(int n=icomp;n<icomp+ncomp;n++)
    FORBOX(gb,i,j,k){
        if(box_contains(vb,i,j,k)) continue;
        int done=0;
        for(int sx=-1;sx<=1&&!done;sx++){
            if(sx!=0&&!g->per[0]) continue;
            for(int sy=-1;sy<=1&&!done;sy++){
                if(sy!=0&&!g->per[1]) continue;
                if(sx==0&&sy==0) continue;
                int ii=i+sx*len[0], jj=j+sy*len[1];
                if(box_contains(vb,ii,jj,k)){ VN(f,i,j,k,n)=VN(f,ii,jj,k,n); done=1; }
            }
        }
    }
}

/* amrex::setBC: domain type where the box reaches the domain side, int_dir otherwise */
static BCRec setbc(BCRec d,Box bx,Box dom)
{ BCRec r; for(int k=0;k<3;k++){ r.lo[k]= bx.lo[k]<=dom.lo[k]?d.lo[k]:BC_INT_DIR; r.hi[k]= bx.hi[k]>=dom.hi[k]?d.hi[k]:BC_INT_DIR; } return r; }
static void need_calc(Fab calc){ if(!calc.p){ fprintf(stderr,"oracle: orlanski_rad side without a calc array (the reference reads an empty Array4 here)\n"); abort(); } }
static const double orl_eps=1.0e-20;
static int copy_like_cons(int t){ return t==BC_FOEXTRAP||t==BC_CLAMPED||t==BC_CHAPMAN||t==BC_ORLANSKI_RAD||t==BC_ORLANSKI_RAD_NUDGE; }
static int skip_corner(int t){ return t==BC_CLAMPED||t==BC_FLATHER||t==BC_CHAPMAN||t==BC_ORLANSKI_RAD_NUDGE; }

/* REMORA_BoundaryConditions_cons.cpp:20-340 */
static void impose_cons(const Geom *g,Fab dest,Box bx,Box valid_bx,Fab mskr,Fab msku,Fab mskv,Fab calc,int icomp,int ncomp,int bccomp,int n_not_fill)
{
    Box dom=g->domain;
    int perx=g->per[0], pery=g->per[1];
    const double eps=orl_eps;
    for(int n=0;n<ncomp;n++){
        BCRec bc=setbc(g->bcs[bccomp+n],bx,dom);                                              /* :33-34 */
        const double *ext=g->extdir[bccomp+n];
        int c=icomp+n;
        int dlx=dom.lo[0], dhx=dom.hi[0], dly=dom.lo[1], dhy=dom.hi[1];
        if(!perx){                                                                            /* :65-107 */
            Box lo=bx; lo.hi[0]=dlx-1;
            lo.lo[1]= valid_bx.lo[1]-1>dly?valid_bx.lo[1]-1:dly; lo.hi[1]= valid_bx.hi[1]+1<dhy?valid_bx.hi[1]+1:dhy;
            Box hi=bx; hi.lo[0]=dhx+1; hi.lo[1]=lo.lo[1]; hi.hi[1]=lo.hi[1];
            if(bc.lo[0]==BC_EXT_DIR) FORBOX(lo,i,j,k) VN(dest,i,j,k,c)=ext[0]*V(mskr,i,j,0);
            else if(bc.lo[0]==BC_ORLANSKI_RAD){ need_calc(calc); FORBOX(lo,i,j,k){                /* :75-87 */
                double grad_lo    =(VN(calc,dlx,j  ,k,c)-VN(calc,dlx,j-1,k,c))*V(mskv,i,j,0);
                double grad_lo_jp1=(VN(calc,dlx,j+1,k,c)-VN(calc,dlx,j  ,k,c))*V(mskv,i,j,0);
                double dTdt=VN(calc,dlx,j,k,c)-VN(dest,dlx  ,j,k,c);
                double dTdx=VN(dest,dlx,j,k,c)-VN(dest,dlx+1,j,k,c);
                if(dTdt*dTdx<0.0) dTdt=0.0;
                double dTde=(dTdt*(grad_lo+grad_lo_jp1)>0.0)?grad_lo:grad_lo_jp1;
                double cff=dmax(dTdx*dTdx+dTde*dTde,eps);
                double Cx=dTdt*dTdx;
                VN(dest,i,j,k,c)=(cff*VN(calc,dlx-1,j,k,c)+Cx*VN(dest,dlx,j,k,c))*V(mskr,i,j,0)/(cff+Cx);
            } }
            if(bc.hi[0]==BC_EXT_DIR) FORBOX(hi,i,j,k) VN(dest,i,j,k,c)=ext[3]*V(mskr,i,j,0);
            else if(bc.hi[0]==BC_ORLANSKI_RAD){ need_calc(calc); FORBOX(hi,i,j,k){                /* :92-104 */
                double grad_hi    =(VN(calc,dhx,j  ,k,c)-VN(calc,dhx,j-1,k,c))*V(mskv,i,j,0);
                double grad_hi_jp1=(VN(calc,dhx,j+1,k,c)-VN(calc,dhx,j  ,k,c))*V(mskv,i,j,0);
                double dTdt=VN(calc,dhx,j,k,c)-VN(dest,dhx  ,j,k,c);
                double dTdx=VN(dest,dhx,j,k,c)-VN(dest,dhx-1,j,k,c);
                if(dTdt*dTdx<0.0) dTdt=0.0;
                double dTde=(dTdt*(grad_hi+grad_hi_jp1)>0.0)?grad_hi:grad_hi_jp1;
                double cff=dmax(dTdx*dTdx+dTde*dTde,eps);
                double Cx=dTdt*dTdx;
                VN(dest,i,j,k,c)=(cff*VN(calc,dhx+1,j,k,c)+Cx*VN(dest,dhx,j,k,c))*V(mskr,i,j,0)/(cff+Cx);
            } }
        }
        if(!pery){                                                                            /* :109-151 */
            Box lo=bx; lo.hi[1]=dly-1;
            lo.lo[0]= valid_bx.lo[0]-1>dlx?valid_bx.lo[0]-1:dlx; lo.hi[0]= valid_bx.hi[0]+1<dhx?valid_bx.hi[0]+1:dhx;
            Box hi=bx; hi.lo[1]=dhy+1; hi.lo[0]=lo.lo[0]; hi.hi[0]=lo.hi[0];
            if(bc.lo[1]==BC_EXT_DIR) FORBOX(lo,i,j,k) VN(dest,i,j,k,c)=ext[1]*V(mskr,i,j,0);
            else if(bc.lo[1]==BC_ORLANSKI_RAD){ need_calc(calc); FORBOX(lo,i,j,k){                /* :119-131 */
                double grad_lo    =(VN(calc,i  ,dly,k,c)-VN(calc,i-1,dly,k,c))*V(msku,i,j,0);
                double grad_lo_ip1=(VN(calc,i+1,dly,k,c)-VN(calc,i  ,dly,k,c))*V(msku,i,j,0);
                double dTdt=VN(calc,i,dly,k,c)-VN(dest,i,dly  ,k,c);
                double dTde=VN(dest,i,dly,k,c)-VN(dest,i,dly+1,k,c);
                if(dTdt*dTde<0.0) dTdt=0.0;
                double dTdx=(dTdt*(grad_lo+grad_lo_ip1)>0.0)?grad_lo:grad_lo_ip1;
                double cff=dmax(dTdx*dTdx+dTde*dTde,eps);
                double Ce=dTdt*dTde;
                VN(dest,i,j,k,c)=(cff*VN(calc,i,dly-1,k,c)+Ce*VN(dest,i,dly,k,c))*V(mskr,i,j,0)/(cff+Ce);
            } }
            if(bc.hi[1]==BC_EXT_DIR) FORBOX(hi,i,j,k) VN(dest,i,j,k,c)=ext[4]*V(mskr,i,j,0);
            else if(bc.hi[1]==BC_ORLANSKI_RAD){ need_calc(calc); FORBOX(hi,i,j,k){                /* :136-148 */
                double grad_hi    =(VN(calc,i  ,dhy,k,c)-VN(calc,i-1,dhy,k,c))*V(msku,i,j,0);
                double grad_hi_ip1=(VN(calc,i+1,dhy,k,c)-VN(calc,i  ,dhy,k,c))*V(msku,i,j,0);
                double dTdt=VN(calc,i,dhy,k,c)-VN(dest,i,dhy  ,k,c);
                double dTde=VN(dest,i,dhy,k,c)-VN(dest,i,dhy-1,k,c);
                if(dTdt*dTde<0.0) dTdt=0.0;
                double dTdx=(dTdt*(grad_hi+grad_hi_ip1)>0.0)?grad_hi:grad_hi_ip1;
                double cff=dmax(dTdx*dTdx+dTde*dTde,eps);
                double Ce=dTdt*dTde;
                VN(dest,i,j,k,c)=(cff*VN(calc,i,dhy+1,k,c)+Ce*VN(dest,i,dhy,k,c))*V(mskr,i,j,0)/(cff+Ce);
            } }
        }
        {                                                                                     /* :153-168 */
            Box lo=bx; lo.hi[2]=dom.lo[2]-1; Box hi=bx; hi.lo[2]=dom.hi[2]+1;
            if(bc.lo[2]==BC_EXT_DIR) FORBOX(lo,i,j,k) VN(dest,i,j,k,c)=ext[2]*V(mskr,i,j,0);
            if(bc.hi[2]==BC_EXT_DIR) FORBOX(hi,i,j,k) VN(dest,i,j,k,c)=ext[5]*V(mskr,i,j,0);
        }
        int zl= dom.lo[2]>bx.lo[2]?dom.lo[2]:bx.lo[2], zh= dom.hi[2]<bx.hi[2]?dom.hi[2]:bx.hi[2];
        Box xlo=bx; xlo.hi[0]=dlx-1-n_not_fill; xlo.lo[2]=zl; xlo.hi[2]=zh;                   /* :170-181 */
        Box xhi=bx; xhi.lo[0]=dhx+1+n_not_fill; xhi.lo[2]=zl; xhi.hi[2]=zh;
        Box ylo=bx; ylo.hi[1]=dly-1-n_not_fill; ylo.lo[2]=zl; ylo.hi[2]=zh;
        Box yhi=bx; yhi.lo[1]=dhy+1+n_not_fill; yhi.lo[2]=zl; yhi.hi[2]=zh;
        Box xlo_ylo=box_and(xlo,ylo), xhi_ylo=box_and(xhi,ylo), xlo_yhi=box_and(xlo,yhi), xhi_yhi=box_and(xhi,yhi);
        if(!perx || bccomp==BCV_FOEXTRAP){                                                    /* :193-221 */
            FORBOX(xlo,i,j,k){
                int iflip=dlx-1-i, inner=(bc.lo[0]==BC_ORLANSKI_RAD)?1:0;
                if(copy_like_cons(bc.lo[0])) VN(dest,i,j,k,c)=VN(dest,dlx-n_not_fill-inner,j,k,c);
                else if(bc.lo[0]==BC_REFLECT_EVEN) VN(dest,i,j,k,c)= VN(dest,iflip,j,k,c);
                else if(bc.lo[0]==BC_REFLECT_ODD)  VN(dest,i,j,k,c)=-VN(dest,iflip,j,k,c);
            }
            FORBOX(xhi,i,j,k){
                int iflip=2*dhx+1-i, inner=(bc.hi[0]==BC_ORLANSKI_RAD)?1:0;
                if(copy_like_cons(bc.hi[0])) VN(dest,i,j,k,c)=VN(dest,dhx+n_not_fill+inner,j,k,c);
                else if(bc.hi[0]==BC_REFLECT_EVEN) VN(dest,i,j,k,c)= VN(dest,iflip,j,k,c);
                else if(bc.hi[0]==BC_REFLECT_ODD)  VN(dest,i,j,k,c)=-VN(dest,iflip,j,k,c);
            }
        }
        if(!pery || bccomp==BCV_FOEXTRAP){                                                    /* :223-252 */
            FORBOX(ylo,i,j,k){
                int jflip=dly-1-j, inner=(bc.lo[1]==BC_ORLANSKI_RAD)?1:0;
                if(copy_like_cons(bc.lo[1])) VN(dest,i,j,k,c)=VN(dest,i,dly-n_not_fill-inner,k,c);
                else if(bc.lo[1]==BC_REFLECT_EVEN) VN(dest,i,j,k,c)= VN(dest,i,jflip,k,c);
                else if(bc.lo[1]==BC_REFLECT_ODD)  VN(dest,i,j,k,c)=-VN(dest,i,jflip,k,c);
            }
            FORBOX(yhi,i,j,k){
                int jflip=2*dhy+1-j, inner=(bc.hi[1]==BC_ORLANSKI_RAD)?1:0;
                /* :242 tests the LOW side's type for orlanski_rad (sic) */
                if(bc.hi[1]==BC_FOEXTRAP||bc.hi[1]==BC_CLAMPED||bc.hi[1]==BC_CHAPMAN||bc.lo[1]==BC_ORLANSKI_RAD||bc.hi[1]==BC_ORLANSKI_RAD_NUDGE)
                    VN(dest,i,j,k,c)=VN(dest,i,dhy+n_not_fill+inner,k,c);
                else if(bc.hi[1]==BC_REFLECT_EVEN) VN(dest,i,j,k,c)= VN(dest,i,jflip,k,c);
                else if(bc.hi[1]==BC_REFLECT_ODD)  VN(dest,i,j,k,c)=-VN(dest,i,jflip,k,c);
            }
        }
        {                                                                                     /* :254-286 (foextrap only: no array on this path has z ghost cells) */
            Box zlo=bx; zlo.hi[2]= dom.lo[2]-1>bx.lo[2]?dom.lo[2]-1:bx.lo[2];
            Box zhi=bx; zhi.lo[2]= dom.hi[2]+1<bx.hi[2]?dom.hi[2]+1:bx.hi[2];
            if(box_ok(zlo)&&bc.lo[2]==BC_FOEXTRAP) FORBOX(zlo,i,j,k) VN(dest,i,j,k,c)=VN(dest,i,j,dom.lo[2],c);
            if(box_ok(zlo)&&bc.hi[2]==BC_FOEXTRAP) FORBOX(zhi,i,j,k) VN(dest,i,j,k,c)=VN(dest,i,j,dom.hi[2],c);
        }
        if((!perx&&!pery) || bccomp==BCV_FOEXTRAP){                                           /* :287-337 */
            if(box_ok(xlo_ylo)&&!skip_corner(bc.lo[0])&&!skip_corner(bc.lo[1])) FORBOX(xlo_ylo,i,j,k) VN(dest,i,j,k,c)=0.5*(VN(dest,i,dly,k,c)+VN(dest,dlx,j,k,c));
            if(box_ok(xlo_yhi)&&!skip_corner(bc.lo[0])&&!skip_corner(bc.hi[1])) FORBOX(xlo_yhi,i,j,k) VN(dest,i,j,k,c)=0.5*(VN(dest,i,dhy,k,c)+VN(dest,dlx,j,k,c));
            if(box_ok(xhi_ylo)&&!skip_corner(bc.hi[0])&&!skip_corner(bc.lo[1])) FORBOX(xhi_ylo,i,j,k) VN(dest,i,j,k,c)=0.5*(VN(dest,i,dly,k,c)+VN(dest,dhx,j,k,c));
            if(box_ok(xhi_yhi)&&!skip_corner(bc.hi[0])&&!skip_corner(bc.hi[1])) FORBOX(xhi_yhi,i,j,k) VN(dest,i,j,k,c)=0.5*(VN(dest,i,dhy,k,c)+VN(dest,dhx,j,k,c));
        }
    }
}

/* REMORA_BoundaryConditions_xvel.cpp:12-269 (dest already offset to icomp; calc is not: PhysBCFunct.cpp:76,80) */
static void impose_xvel(const Geom *g,Fab dest,Box bx,Fab msku,Fab calc,int bccomp)
{
    Box dom=g->domain; int perx=g->per[0], pery=g->per[1];
    BCRec bc=setbc(g->bcs[bccomp],bx,dom); const double *ext=g->extdir[bccomp];
    const double eps=orl_eps;
    int dlx=dom.lo[0], dhx=dom.hi[0], dly=dom.lo[1], dhy=dom.hi[1];
    int fx=(bccomp==BCV_FOEXTRAP);
    if(!perx||fx){                                                                            /* :56-134 */
        Box xlo=bx; xlo.hi[0]=dlx-1; Box xhi=bx; xhi.lo[0]=dhx+2;
        Box flo=bx; flo.lo[0]=flo.hi[0]=dlx; Box fhi=bx; fhi.lo[0]=fhi.hi[0]=dhx+1;
        xlo=box_grow(xlo,0,-1,0); xhi=box_grow(xhi,0,-1,0); flo=box_grow(flo,0,-1,0); fhi=box_grow(fhi,0,-1,0);
        FORBOX(flo,i,j,k){
            if(bc.lo[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[0]*V(msku,i,j,0);
            else if(bc.lo[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dlx+1,j,k)*V(msku,i,j,0);
            else if(bc.lo[0]==BC_ORLANSKI_RAD){                                               /* :69-81 */
                need_calc(calc);
                double grad_lo_ip1 =V(calc,dlx+1,j  ,k)-V(calc,dlx+1,j-1,k);
                double grad_lo_ijp1=V(calc,dlx+1,j+1,k)-V(calc,dlx+1,j  ,k);
                double dUdt=V(calc,dlx+1,j,k)-V(dest,dlx+1,j,k);
                double dUdx=V(dest,dlx+1,j,k)-V(dest,dlx+2,j,k);
                if(dUdt*dUdx<0.0) dUdt=0.0;
                double dUde=(dUdt*(grad_lo_ip1+grad_lo_ijp1)>0.0)?grad_lo_ip1:grad_lo_ijp1;
                double cff=dmax(dUdx*dUdx+dUde*dUde,eps);
                double Cx=dUdt*dUdx;
                V(dest,i,j,k)=(cff*V(calc,i,j,k)+Cx*V(dest,dlx+1,j,k))*V(msku,i,j,0)/(cff+Cx);
            }
        }
        FORBOX(xlo,i,j,k){
            int inner=(bc.lo[0]==BC_FOEXTRAP)?1:0, iflip=dlx-i;
            if(bc.lo[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[0]*V(msku,i,j,0);
            else if(bc.lo[0]==BC_FOEXTRAP||bc.lo[0]==BC_CLAMPED||bc.lo[0]==BC_ORLANSKI_RAD) V(dest,i,j,k)=V(dest,dlx+inner,j,k)*V(msku,i,j,0);
            else if(bc.lo[0]==BC_REFLECT_EVEN) V(dest,i,j,k)= V(dest,iflip,j,k)*V(msku,i,j,0);
            else if(bc.lo[0]==BC_REFLECT_ODD)  V(dest,i,j,k)=-V(dest,iflip,j,k)*V(msku,i,j,0);
        }
        FORBOX(fhi,i,j,k){
            if(bc.hi[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[3]*V(msku,i,j,0);
            else if(bc.hi[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dhx,j,k)*V(msku,i,j,0);
            else if(bc.hi[0]==BC_ORLANSKI_RAD){                                               /* :105-117 */
                need_calc(calc);
                double grad_hi    =V(calc,dhx,j  ,k)-V(calc,dhx,j-1,k);
                double grad_hi_jp1=V(calc,dhx,j+1,k)-V(calc,dhx,j  ,k);
                double dUdt=V(calc,dhx,j,k)-V(dest,dhx  ,j,k);
                double dUdx=V(dest,dhx,j,k)-V(dest,dhx-1,j,k);
                if(dUdt*dUdx<0.0) dUdt=0.0;
                double dUde=(dUdt*(grad_hi+grad_hi_jp1)>0.0)?grad_hi:grad_hi_jp1;
                double cff=dmax(dUdx*dUdx+dUde*dUde,eps);
                double Cx=dUdt*dUdx;
                V(dest,i,j,k)=(cff*V(calc,dhx+1,j,k)+Cx*V(dest,dhx,j,k))*V(msku,i,j,0)/(cff+Cx);
            }
        }
        FORBOX(xhi,i,j,k){
            int inner=(bc.hi[0]==BC_FOEXTRAP)?1:0, iflip=2*(dhx+1)-i;
            if(bc.hi[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[3]*V(msku,i,j,0);
            else if(bc.hi[0]==BC_FOEXTRAP||bc.hi[0]==BC_CLAMPED||bc.hi[0]==BC_ORLANSKI_RAD) V(dest,i,j,k)=V(dest,dhx+1-inner,j,k)*V(msku,i,j,0);
            else if(bc.hi[0]==BC_REFLECT_EVEN) V(dest,i,j,k)= V(dest,iflip,j,k)*V(msku,i,j,0);
            else if(bc.hi[0]==BC_REFLECT_ODD)  V(dest,i,j,k)=-V(dest,iflip,j,k)*V(msku,i,j,0);
        }
    }
    if(!pery||fx){                                                                            /* :136-191 */
        Box ylo=bx; ylo.hi[1]=dly-1; Box yhi=bx; yhi.lo[1]=dhy+1;
        ylo=box_grow(ylo,-1,0,0); yhi=box_grow(yhi,-1,0,0);
        FORBOX(ylo,i,j,k){
            int jflip=dly-1-j;
            if(bc.lo[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[1]*V(msku,i,j,0);
            else if(bc.lo[1]==BC_FOEXTRAP||bc.lo[1]==BC_CLAMPED) V(dest,i,j,k)=V(dest,i,dly,k)*V(msku,i,j,0);
            else if(bc.lo[1]==BC_ORLANSKI_RAD){                                               /* :148-159 */
                need_calc(calc);
                double grad_lo    =V(calc,i+1,dly,k)-V(calc,i  ,dly,k);
                double grad_lo_im1=V(calc,i  ,dly,k)-V(calc,i-1,dly,k);
                double dUdt=V(calc,i,dly,k)-V(dest,i,dly  ,k);
                double dUde=V(dest,i,dly,k)-V(dest,i,dly+1,k);
                if(dUdt*dUde<0.0) dUdt=0.0;
                double dUdx=(dUdt*(grad_lo_im1+grad_lo)>0.0)?grad_lo_im1:grad_lo;
                double cff=dmax(dUdx*dUdx+dUde*dUde,eps);
                double Ce=dUdt*dUde;
                V(dest,i,j,k)=(cff*V(calc,i,dly-1,k)+Ce*V(dest,i,dly,k))*V(msku,i,j,0)/(cff+Ce);
            }
            else if(bc.lo[1]==BC_REFLECT_EVEN) V(dest,i,j,k)= V(dest,i,jflip,k)*V(msku,i,j,0);
            else if(bc.lo[1]==BC_REFLECT_ODD)  V(dest,i,j,k)=-V(dest,i,jflip,k)*V(msku,i,j,0);
        }
        FORBOX(yhi,i,j,k){
            int jflip=2*dhy+1-j;
            if(bc.hi[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[4]*V(msku,i,j,0);
            else if(bc.hi[1]==BC_FOEXTRAP||bc.hi[1]==BC_CLAMPED) V(dest,i,j,k)=V(dest,i,dhy,k)*V(msku,i,j,0);
            else if(bc.hi[1]==BC_ORLANSKI_RAD){                                               /* :172-183 */
                need_calc(calc);
                double grad_hi    =V(calc,i+1,dhy,k)-V(calc,i  ,dhy,k);
                double grad_hi_im1=V(calc,i  ,dhy,k)-V(calc,i-1,dhy,k);
                double dUdt=V(calc,i,dhy,k)-V(dest,i,dhy  ,k);
                double dUde=V(dest,i,dhy,k)-V(dest,i,dhy-1,k);
                if(dUdt*dUde<0.0) dUdt=0.0;
                double dUdx=(dUdt*(grad_hi_im1+grad_hi)>0.0)?grad_hi_im1:grad_hi;
                double cff=dmax(dUdx*dUdx+dUde*dUde,eps);
                double Ce=dUdt*dUde;
                V(dest,i,j,k)=(cff*V(calc,i,dhy+1,k)+Ce*V(dest,i,dhy,k))*V(msku,i,j,0)/(cff+Ce);
            }
            else if(bc.hi[1]==BC_REFLECT_EVEN) V(dest,i,j,k)= V(dest,i,jflip,k)*V(msku,i,j,0);
            else if(bc.hi[1]==BC_REFLECT_ODD)  V(dest,i,j,k)=-V(dest,i,jflip,k)*V(msku,i,j,0);
        }
    }
    {                                                                                         /* :193-223 */
        Box zlo=bx; zlo.hi[2]=dom.lo[2]-1; Box zhi=bx; zhi.lo[2]=dom.hi[2]+1;
        FORBOX(zlo,i,j,k){ if(bc.lo[2]==BC_EXT_DIR) V(dest,i,j,k)=ext[2]*V(msku,i,j,0); else if(bc.lo[2]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,j,dom.lo[2])*V(msku,i,j,0); }
        FORBOX(zhi,i,j,k){ if(bc.hi[2]==BC_EXT_DIR) V(dest,i,j,k)=ext[5]*V(msku,i,j,0); else if(bc.hi[2]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,j,dom.hi[2])*V(msku,i,j,0); }
    }
    if((!perx||fx)&&(!pery||fx)){                                                             /* :225-265 */
        Box xlo=bx; xlo.hi[0]=dlx; Box xhi=bx; xhi.lo[0]=dhx+1;
        Box ylo=bx; ylo.hi[1]=dly-1; Box yhi=bx; yhi.lo[1]=dhy+1;
        Box a=box_and(xlo,ylo), b=box_and(xlo,yhi), c=box_and(xhi,ylo), d=box_and(xhi,yhi);
        /* :236-239 test the DOMAIN's types, not the box's */
        int cw=g->bcs[bccomp].lo[0]==BC_CLAMPED, ce=g->bcs[bccomp].hi[0]==BC_CLAMPED, cs=g->bcs[bccomp].lo[1]==BC_CLAMPED, cn=g->bcs[bccomp].hi[1]==BC_CLAMPED;
        if(!cw&&!cs) FORBOX(a,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dly,k)+V(dest,dlx+1,j,k));
        if(!cw&&!cn) FORBOX(b,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dhy,k)+V(dest,dlx+1,j,k));
        if(!ce&&!cs) FORBOX(c,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dly,k)+V(dest,dhx,j,k));
        if(!ce&&!cn) FORBOX(d,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dhy,k)+V(dest,dhx,j,k));
    }
}

/* REMORA_BoundaryConditions_yvel.cpp:12-268 */
static void impose_yvel(const Geom *g,Fab dest,Box bx,Fab mskv,Fab calc,int bccomp)
{
    Box dom=g->domain; int perx=g->per[0], pery=g->per[1];
    BCRec bc=setbc(g->bcs[bccomp],bx,dom); const double *ext=g->extdir[bccomp];
    const double eps=orl_eps;
    int dlx=dom.lo[0], dhx=dom.hi[0], dly=dom.lo[1], dhy=dom.hi[1];
    int fx=(bccomp==BCV_FOEXTRAP);
    if(!perx||fx){                                                                            /* :55-110 */
        Box xlo=bx; xlo.hi[0]=dlx-1; Box xhi=bx; xhi.lo[0]=dhx+1;
        xlo=box_grow(xlo,0,-1,0); xhi=box_grow(xhi,0,-1,0);
        FORBOX(xlo,i,j,k){
            int iflip=dlx-1-i;
            if(bc.lo[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[0]*V(mskv,i,j,0);
            else if(bc.lo[0]==BC_FOEXTRAP||bc.lo[0]==BC_CLAMPED) V(dest,i,j,k)=V(dest,dlx,j,k)*V(mskv,i,j,0);
            else if(bc.lo[0]==BC_ORLANSKI_RAD){                                               /* :67-78 */
                need_calc(calc);
                double grad_lo    =V(calc,dlx,j+1,k)-V(calc,dlx,j  ,k);
                double grad_lo_jm1=V(calc,dlx,j  ,k)-V(calc,dlx,j-1,k);
                double dVdt=V(calc,dlx,j,k)-V(dest,dlx  ,j,k);
                double dVdx=V(dest,dlx,j,k)-V(dest,dlx+1,j,k);
                if(dVdt*dVdx<0.0) dVdt=0.0;
                double dVde=(dVdt*(grad_lo_jm1+grad_lo)>0.0)?grad_lo_jm1:grad_lo;
                double cff=dmax(dVdx*dVdx+dVde*dVde,eps);
                double Cx=dVdt*dVdx;
                V(dest,i,j,k)=(cff*V(calc,dlx-1,j,k)+Cx*V(dest,dlx,j,k))*V(mskv,i,j,0)/(cff+Cx);
            }
            else if(bc.lo[0]==BC_REFLECT_EVEN) V(dest,i,j,k)= V(dest,iflip,j,k)*V(mskv,i,j,0);
            else if(bc.lo[0]==BC_REFLECT_ODD)  V(dest,i,j,k)=-V(dest,iflip,j,k)*V(mskv,i,j,0);
        }
        FORBOX(xhi,i,j,k){
            int iflip=2*dhx+1-i;
            if(bc.hi[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[3]*V(mskv,i,j,0);
            else if(bc.hi[0]==BC_FOEXTRAP||bc.hi[0]==BC_CLAMPED) V(dest,i,j,k)=V(dest,dhx,j,k)*V(mskv,i,j,0);
            else if(bc.hi[0]==BC_ORLANSKI_RAD){                                               /* :91-102 */
                need_calc(calc);
                double grad_hi    =V(calc,dhx,j+1,k)-V(calc,dhx,j  ,k);
                double grad_hi_jm1=V(calc,dhx,j  ,k)-V(calc,dhx,j-1,k);
                double dVdt=V(calc,dhx,j,k)-V(dest,dhx  ,j,k);
                double dVdx=V(dest,dhx,j,k)-V(dest,dhx-1,j,k);
                if(dVdt*dVdx<0.0) dVdt=0.0;
                double dVde=(dVdt*(grad_hi_jm1+grad_hi)>0.0)?grad_hi_jm1:grad_hi;
                double cff=dmax(dVdx*dVdx+dVde*dVde,eps);
                double Cx=dVdt*dVdx;
                V(dest,i,j,k)=(cff*V(calc,dhx+1,j,k)+Cx*V(dest,dhx,j,k))*V(mskv,i,j,0)/(cff+Cx);
            }
            else if(bc.hi[0]==BC_REFLECT_EVEN) V(dest,i,j,k)= V(dest,iflip,j,k)*V(mskv,i,j,0);
            else if(bc.hi[0]==BC_REFLECT_ODD)  V(dest,i,j,k)=-V(dest,iflip,j,k)*V(mskv,i,j,0);
        }
    }
    if(!pery||fx){                                                                            /* :112-192 */
        Box ylo=bx; ylo.hi[1]=dly-1; Box yhi=bx; yhi.lo[1]=dhy+2;
        Box flo=bx; flo.lo[1]=flo.hi[1]=dly; Box fhi=bx; fhi.lo[1]=fhi.hi[1]=dhy+1;
        ylo=box_grow(ylo,-1,0,0); yhi=box_grow(yhi,-1,0,0); flo=box_grow(flo,-1,0,0); fhi=box_grow(fhi,-1,0,0);
        FORBOX(flo,i,j,k){
            if(bc.lo[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[1]*V(mskv,i,j,0);
            else if(bc.lo[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dly+1,k)*V(mskv,i,j,0);
            else if(bc.lo[1]==BC_ORLANSKI_RAD){                                               /* :127-138 */
                need_calc(calc);
                double grad_lo_jp1 =V(calc,i  ,dly+1,k)-V(calc,i-1,dly+1,k);
                double grad_lo_ijp1=V(calc,i+1,dly+1,k)-V(calc,i  ,dly+1,k);
                double dVdt=V(calc,i,dly+1,k)-V(dest,i,dly+1,k);
                double dVde=V(dest,i,dly+1,k)-V(dest,i,dly+2,k);
                if(dVdt*dVde<0.0) dVdt=0.0;
                double dVdx=(dVdt*(grad_lo_jp1+grad_lo_ijp1)>0.0)?grad_lo_jp1:grad_lo_ijp1;
                double cff=dmax(dVdx*dVdx+dVde*dVde,eps);
                double Ce=dVdt*dVde;
                V(dest,i,j,k)=(cff*V(calc,i,dly,k)+Ce*V(dest,i,dly+1,k))*V(mskv,i,j,0)/(cff+Ce);
            }
        }
        FORBOX(ylo,i,j,k){
            int inner=(bc.lo[1]==BC_FOEXTRAP)?1:0, jflip=dly-j;
            if(bc.lo[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[1]*V(mskv,i,j,0);
            else if(bc.lo[1]==BC_FOEXTRAP||bc.lo[1]==BC_CLAMPED||bc.lo[1]==BC_ORLANSKI_RAD) V(dest,i,j,k)=V(dest,i,dly+inner,k)*V(mskv,i,j,0);
            else if(bc.lo[1]==BC_REFLECT_EVEN) V(dest,i,j,k)= V(dest,i,jflip,k)*V(mskv,i,j,0);
            else if(bc.lo[1]==BC_REFLECT_ODD)  V(dest,i,j,k)=-V(dest,i,jflip,k)*V(mskv,i,j,0);
        }
        FORBOX(fhi,i,j,k){
            if(bc.hi[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[4]*V(mskv,i,j,0);
            else if(bc.hi[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dhy,k)*V(mskv,i,j,0);
            else if(bc.hi[1]==BC_ORLANSKI_RAD){                                               /* :163-175 */
                need_calc(calc);
                double grad_hi    =V(calc,i  ,dhy,k)-V(calc,i-1,dhy,k);
                double grad_hi_ip1=V(calc,i+1,dhy,k)-V(calc,i  ,dhy,k);
                double dVdt=V(calc,i,dhy,k)-V(dest,i,dhy  ,k);
                double dVde=V(dest,i,dhy,k)-V(dest,i,dhy-1,k);
                if(dVdt*dVde<0.0) dVdt=0.0;
                double dVdx=(dVdt*(grad_hi+grad_hi_ip1)>0.0)?grad_hi:grad_hi_ip1;
                double cff=dmax(dVdx*dVdx+dVde*dVde,eps);
                double Ce=dVdt*dVde;
                V(dest,i,j,k)=(cff*V(calc,i,dhy+1,k)+Ce*V(dest,i,dhy,k))*V(mskv,i,j,0)/(cff+Ce);
            }
        }
        FORBOX(yhi,i,j,k){
            int inner=(bc.hi[1]==BC_FOEXTRAP)?1:0, jflip=2*(dhy+1)-j;
            if(bc.hi[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[4]*V(mskv,i,j,0);
            else if(bc.hi[1]==BC_FOEXTRAP||bc.hi[1]==BC_CLAMPED||bc.hi[1]==BC_ORLANSKI_RAD) V(dest,i,j,k)=V(dest,i,dhy+1-inner,k)*V(mskv,i,j,0);
            else if(bc.hi[1]==BC_REFLECT_EVEN) V(dest,i,j,k)= V(dest,i,jflip,k)*V(mskv,i,j,0);
            else if(bc.hi[1]==BC_REFLECT_ODD)  V(dest,i,j,k)=-V(dest,i,jflip,k)*V(mskv,i,j,0);
        }
    }
    {                                                                                         /* :194-224 */
        Box zlo=bx; zlo.hi[2]=dom.lo[2]-1; Box zhi=bx; zhi.lo[2]=dom.hi[2]+1;
        FORBOX(zlo,i,j,k){ if(bc.lo[2]==BC_EXT_DIR) V(dest,i,j,k)=ext[2]*V(mskv,i,j,0); else if(bc.lo[2]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,j,dom.lo[2])*V(mskv,i,j,0); }
        FORBOX(zhi,i,j,k){ if(bc.hi[2]==BC_EXT_DIR) V(dest,i,j,k)=ext[5]*V(mskv,i,j,0); else if(bc.hi[2]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,j,dom.hi[2])*V(mskv,i,j,0); }
    }
    if((!perx||fx)&&(!pery||fx)){                                                             /* :226-265 */
        Box xlo=bx; xlo.hi[0]=dlx-1; Box xhi=bx; xhi.lo[0]=dhx+1;
        Box ylo=bx; ylo.hi[1]=dly; Box yhi=bx; yhi.lo[1]=dhy+1;
        Box a=box_and(xlo,ylo), b=box_and(xlo,yhi), c=box_and(xhi,ylo), d=box_and(xhi,yhi);
        int cw=g->bcs[bccomp].lo[0]==BC_CLAMPED, ce=g->bcs[bccomp].hi[0]==BC_CLAMPED, cs=g->bcs[bccomp].lo[1]==BC_CLAMPED, cn=g->bcs[bccomp].hi[1]==BC_CLAMPED;
        if(!cw&&!cs) FORBOX(a,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dly+1,k)+V(dest,dlx,j,k));
        if(!cw&&!cn) FORBOX(b,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dhy,k)+V(dest,dlx,j,k));
        if(!ce&&!cs) FORBOX(c,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dly+1,k)+V(dest,dhx,j,k));
        if(!ce&&!cn) FORBOX(d,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dhy,k)+V(dest,dhx,j,k));
    }
}

/* REMORA_BoundaryConditions_zvel.cpp:13-124 */
static void impose_zvel(const Geom *g,Fab dest,Box bx,int bccomp)
{
    Box dom=g->domain; int perx=g->per[0], pery=g->per[1];
    BCRec bc=setbc(g->bcs[bccomp],bx,dom); const double *ext=g->extdir[bccomp];
    int fx=(bccomp==BCV_FOEXTRAP);
    if(!perx||fx){
        Box xlo=bx; xlo.hi[0]=dom.lo[0]-1; Box xhi=bx; xhi.lo[0]=dom.hi[0]+1;
        FORBOX(xlo,i,j,k){ if(bc.lo[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[0]; else if(bc.lo[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dom.lo[0],j,k); }
        FORBOX(xhi,i,j,k){ if(bc.hi[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[3]; else if(bc.hi[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dom.hi[0],j,k); }
    }
    if(!pery||fx){
        Box ylo=bx; ylo.hi[1]=dom.lo[1]-1; Box yhi=bx; yhi.lo[1]=dom.hi[1]+1;
        FORBOX(ylo,i,j,k){ if(bc.lo[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[1]; else if(bc.lo[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dom.lo[1],k); }
        FORBOX(yhi,i,j,k){ if(bc.hi[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[4]; else if(bc.hi[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dom.hi[1],k); }
    }
}

/* REMORA_PhysBCFunct.cpp:21-93 */
void o_physbc_calc(const Geom *g,MF *mf,Fab msk,int icomp,int ncomp,int bccomp,int n_not_fill,Fab calc,Fab msku,Fab mskv)
{
    Box gdom=g->domain;
    for(int d=0;d<3;d++) if(mf->ix[d]) gdom.hi[d]+=1;
    /* 2-D arrays live on a k=0 slab of the domain */
    for(int d=0;d<2;d++) if(g->per[d] && bccomp!=BCV_FOEXTRAP){ gdom.lo[d]-=mf->ng[d]; gdom.hi[d]+=mf->ng[d]; }
    Box bx=fab_box(mf->f);
    if(box_contains_box(gdom,bx)) return;
    if(!mf->ix[0]&&!mf->ix[1]&&!mf->ix[2]) impose_cons(g,mf->f,bx,mf->valid,msk,msku,mskv,calc,icomp,ncomp,bccomp,n_not_fill);
    else if(mf->ix[0]) impose_xvel(g,fab_comp(mf->f,icomp),bx,msk,calc,bccomp);
    else if(mf->ix[1]) impose_yvel(g,fab_comp(mf->f,icomp),bx,msk,calc,bccomp);
    else if(mf->ix[2]) impose_zvel(g,fab_comp(mf->f,icomp),bx,bccomp);
}
static Fab null_fab(void){ Fab f; memset(&f,0,sizeof f); return f; }
void o_physbc(const Geom *g,MF *mf,Fab msk,int icomp,int ncomp,int bccomp,int n_not_fill)
{ o_physbc_calc(g,mf,msk,icomp,ncomp,bccomp,n_not_fill,null_fab(),null_fab(),null_fab()); }

void o_fillpatch_calc(const Geom *g,MF *mf,Fab msk,int bccomp,int icomp,int ncomp,Fab calc,Fab msku,Fab mskv)
{
    o_fill_boundary(g,mf,icomp,ncomp);
    o_physbc_calc(g,mf,msk,icomp,ncomp,bccomp,0,calc,msku,mskv);
}
void o_fillpatch(const Geom *g,MF *mf,Fab msk,int bccomp,int icomp,int ncomp)
{ o_fillpatch_calc(g,mf,msk,bccomp,icomp,ncomp,null_fab(),null_fab(),null_fab()); }
