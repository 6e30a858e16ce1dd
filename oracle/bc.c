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

static void bc_setall(BCRec *b,int dir,int v){ b->lo[dir]=v; b->hi[dir]=v; }

void o_init_bcs(Geom *g)
{
    const Params *P=g->prm;
    for(int v=0;v<BCV_NUM;v++) for(int d=0;d<3;d++){ g->bcs[v].lo[d]=BC_BOGUS; g->bcs[v].hi[d]=BC_BOGUS; }
    for(int v=0;v<BCV_NUM;v++) for(int o=0;o<6;o++) g->extdir[v][o]=0.0;
    for(int o=0;o<6;o++){ g->extdir[BCV_TEMP][o]=1.e19; g->extdir[BCV_SALT][o]=1.e20; g->extdir[BCV_SCALAR][o]=1.e21; }   /* :161-163 */
    for(int dir=0;dir<3;dir++){
        int per = (dir<2) ? P->periodic[dir] : 0;     /* z always SlipWall (:130) */
        int base = per ? BC_INT_DIR : BC_FOEXTRAP;
        /* xvel,yvel (:218-309): tangential foextrap, normal component ext_dir; periodic -> int_dir */
        bc_setall(&g->bcs[BCV_XVEL],dir,base); bc_setall(&g->bcs[BCV_YVEL],dir,base);
        if(!per && dir<2) bc_setall(&g->bcs[BCV_XVEL+dir],dir,BC_EXT_DIR);
        bc_setall(&g->bcs[BCV_ZVEL],dir,BC_FOEXTRAP);                                     /* :311-316 */
        for(int n=0;n<NCONS;n++) bc_setall(&g->bcs[BCV_CONS+n],dir,base);                 /* :331-409 */
        bc_setall(&g->bcs[BCV_UBAR],dir,base); bc_setall(&g->bcs[BCV_VBAR],dir,base);     /* :425-541 */
        bc_setall(&g->bcs[BCV_U2D_SIMPLE],dir,base); bc_setall(&g->bcs[BCV_V2D_SIMPLE],dir,base);
        if(!per && dir<2){ bc_setall(&g->bcs[BCV_UBAR+dir],dir,BC_EXT_DIR); bc_setall(&g->bcs[BCV_U2D_SIMPLE+dir],dir,BC_EXT_DIR); }
        bc_setall(&g->bcs[BCV_ZETA],dir,base); bc_setall(&g->bcs[BCV_TKE],dir,base);      /* :556-627 */
        bc_setall(&g->bcs[BCV_FOEXTRAP_PERIODIC],dir,BC_FOEXTRAP);                        /* :636-648 */
        bc_setall(&g->bcs[BCV_FOEXTRAP],dir,BC_FOEXTRAP);                                 /* :655-667 */
    }
}

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
static void unsupported(int t){ if(t!=BC_INT_DIR&&t!=BC_FOEXTRAP&&t!=BC_EXT_DIR&&t!=BC_BOGUS){ fprintf(stderr,"oracle: BC type %d not restated\n",t); abort(); } }

/* REMORA_BoundaryConditions_cons.cpp:20-340 */
static void impose_cons(const Geom *g,Fab dest,Box bx,Box valid_bx,Fab mskr,int icomp,int ncomp,int bccomp,int n_not_fill)
{
    Box dom=g->domain;
    int perx=g->per[0], pery=g->per[1];
    for(int n=0;n<ncomp;n++){
        BCRec bc=setbc(g->bcs[bccomp+n],bx,dom);                                              /* :33-34 */
        for(int d=0;d<3;d++){ unsupported(bc.lo[d]); unsupported(bc.hi[d]); }
        const double *ext=g->extdir[bccomp+n];
        int c=icomp+n;
        if(!perx){                                                                            /* :65-107 */
            Box lo=bx; lo.hi[0]=dom.lo[0]-1;
            lo.lo[1]= valid_bx.lo[1]-1>dom.lo[1]?valid_bx.lo[1]-1:dom.lo[1]; lo.hi[1]= valid_bx.hi[1]+1<dom.hi[1]?valid_bx.hi[1]+1:dom.hi[1];
            Box hi=bx; hi.lo[0]=dom.hi[0]+1; hi.lo[1]=lo.lo[1]; hi.hi[1]=lo.hi[1];
            if(bc.lo[0]==BC_EXT_DIR) FORBOX(lo,i,j,k) VN(dest,i,j,k,c)=ext[0]*V(mskr,i,j,0);
            if(bc.hi[0]==BC_EXT_DIR) FORBOX(hi,i,j,k) VN(dest,i,j,k,c)=ext[3]*V(mskr,i,j,0);
        }
        if(!pery){                                                                            /* :109-151 */
            Box lo=bx; lo.hi[1]=dom.lo[1]-1;
            lo.lo[0]= valid_bx.lo[0]-1>dom.lo[0]?valid_bx.lo[0]-1:dom.lo[0]; lo.hi[0]= valid_bx.hi[0]+1<dom.hi[0]?valid_bx.hi[0]+1:dom.hi[0];
            Box hi=bx; hi.lo[1]=dom.hi[1]+1; hi.lo[0]=lo.lo[0]; hi.hi[0]=lo.hi[0];
            if(bc.lo[1]==BC_EXT_DIR) FORBOX(lo,i,j,k) VN(dest,i,j,k,c)=ext[1]*V(mskr,i,j,0);
            if(bc.hi[1]==BC_EXT_DIR) FORBOX(hi,i,j,k) VN(dest,i,j,k,c)=ext[4]*V(mskr,i,j,0);
        }
        {                                                                                     /* :153-168 */
            Box lo=bx; lo.hi[2]=dom.lo[2]-1; Box hi=bx; hi.lo[2]=dom.hi[2]+1;
            if(bc.lo[2]==BC_EXT_DIR) FORBOX(lo,i,j,k) VN(dest,i,j,k,c)=ext[2]*V(mskr,i,j,0);
            if(bc.hi[2]==BC_EXT_DIR) FORBOX(hi,i,j,k) VN(dest,i,j,k,c)=ext[5]*V(mskr,i,j,0);
        }
        int zl= dom.lo[2]>bx.lo[2]?dom.lo[2]:bx.lo[2], zh= dom.hi[2]<bx.hi[2]?dom.hi[2]:bx.hi[2];
        Box xlo=bx; xlo.hi[0]=dom.lo[0]-1-n_not_fill; xlo.lo[2]=zl; xlo.hi[2]=zh;             /* :170-181 */
        Box xhi=bx; xhi.lo[0]=dom.hi[0]+1+n_not_fill; xhi.lo[2]=zl; xhi.hi[2]=zh;
        Box ylo=bx; ylo.hi[1]=dom.lo[1]-1-n_not_fill; ylo.lo[2]=zl; ylo.hi[2]=zh;
        Box yhi=bx; yhi.lo[1]=dom.hi[1]+1+n_not_fill; yhi.lo[2]=zl; yhi.hi[2]=zh;
        Box xlo_ylo=box_and(xlo,ylo), xhi_ylo=box_and(xhi,ylo), xlo_yhi=box_and(xlo,yhi), xhi_yhi=box_and(xhi,yhi);
        if(!perx || bccomp==BCV_FOEXTRAP){                                                    /* :193-221 */
            if(bc.lo[0]==BC_FOEXTRAP) FORBOX(xlo,i,j,k) VN(dest,i,j,k,c)=VN(dest,dom.lo[0]-n_not_fill,j,k,c);
            if(bc.hi[0]==BC_FOEXTRAP) FORBOX(xhi,i,j,k) VN(dest,i,j,k,c)=VN(dest,dom.hi[0]+n_not_fill,j,k,c);
        }
        if(!pery || bccomp==BCV_FOEXTRAP){                                                    /* :223-252 */
            if(bc.lo[1]==BC_FOEXTRAP) FORBOX(ylo,i,j,k) VN(dest,i,j,k,c)=VN(dest,i,dom.lo[1]-n_not_fill,k,c);
            if(bc.hi[1]==BC_FOEXTRAP) FORBOX(yhi,i,j,k) VN(dest,i,j,k,c)=VN(dest,i,dom.hi[1]+n_not_fill,k,c);
        }
        {                                                                                     /* :254-286 */
            Box zlo=bx; zlo.hi[2]= dom.lo[2]-1>bx.lo[2]?dom.lo[2]-1:bx.lo[2];
            Box zhi=bx; zhi.lo[2]= dom.hi[2]+1<bx.hi[2]?dom.hi[2]+1:bx.hi[2];
            if(box_ok(zlo)&&bc.lo[2]==BC_FOEXTRAP) FORBOX(zlo,i,j,k) VN(dest,i,j,k,c)=VN(dest,i,j,dom.lo[2],c);
            if(box_ok(zlo)&&bc.hi[2]==BC_FOEXTRAP) FORBOX(zhi,i,j,k) VN(dest,i,j,k,c)=VN(dest,i,j,dom.hi[2],c);
        }
        if((!perx&&!pery) || bccomp==BCV_FOEXTRAP){                                           /* :287-337 */
            if(box_ok(xlo_ylo)) FORBOX(xlo_ylo,i,j,k) VN(dest,i,j,k,c)=0.5*(VN(dest,i,dom.lo[1],k,c)+VN(dest,dom.lo[0],j,k,c));
            if(box_ok(xlo_yhi)) FORBOX(xlo_yhi,i,j,k) VN(dest,i,j,k,c)=0.5*(VN(dest,i,dom.hi[1],k,c)+VN(dest,dom.lo[0],j,k,c));
            if(box_ok(xhi_ylo)) FORBOX(xhi_ylo,i,j,k) VN(dest,i,j,k,c)=0.5*(VN(dest,i,dom.lo[1],k,c)+VN(dest,dom.hi[0],j,k,c));
            if(box_ok(xhi_yhi)) FORBOX(xhi_yhi,i,j,k) VN(dest,i,j,k,c)=0.5*(VN(dest,i,dom.hi[1],k,c)+VN(dest,dom.hi[0],j,k,c));
        }
    }
}

/* REMORA_BoundaryConditions_xvel.cpp:12-269 (dest already offset to icomp) */
static void impose_xvel(const Geom *g,Fab dest,Box bx,Fab msku,int bccomp)
{
    Box dom=g->domain; int perx=g->per[0], pery=g->per[1];
    BCRec bc=setbc(g->bcs[bccomp],bx,dom); const double *ext=g->extdir[bccomp];
    for(int d=0;d<3;d++){ unsupported(bc.lo[d]); unsupported(bc.hi[d]); }
    int fx=(bccomp==BCV_FOEXTRAP);
    if(!perx||fx){                                                                            /* :56-134 */
        Box xlo=bx; xlo.hi[0]=dom.lo[0]-1; Box xhi=bx; xhi.lo[0]=dom.hi[0]+2;
        Box flo=bx; flo.lo[0]=flo.hi[0]=dom.lo[0]; Box fhi=bx; fhi.lo[0]=fhi.hi[0]=dom.hi[0]+1;
        xlo=box_grow(xlo,0,-1,0); xhi=box_grow(xhi,0,-1,0); flo=box_grow(flo,0,-1,0); fhi=box_grow(fhi,0,-1,0);
        FORBOX(flo,i,j,k){
            if(bc.lo[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[0]*V(msku,i,j,0);
            else if(bc.lo[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dom.lo[0]+1,j,k)*V(msku,i,j,0);
        }
        FORBOX(xlo,i,j,k){
            int inner=(bc.lo[0]==BC_FOEXTRAP)?1:0;
            if(bc.lo[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[0]*V(msku,i,j,0);
            else if(bc.lo[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dom.lo[0]+inner,j,k)*V(msku,i,j,0);
        }
        FORBOX(fhi,i,j,k){
            if(bc.hi[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[3]*V(msku,i,j,0);
            else if(bc.hi[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dom.hi[0],j,k)*V(msku,i,j,0);
        }
        FORBOX(xhi,i,j,k){
            int inner=(bc.hi[0]==BC_FOEXTRAP)?1:0;
            if(bc.hi[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[3]*V(msku,i,j,0);
            else if(bc.hi[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dom.hi[0]+1-inner,j,k)*V(msku,i,j,0);
        }
    }
    if(!pery||fx){                                                                            /* :136-191 */
        Box ylo=bx; ylo.hi[1]=dom.lo[1]-1; Box yhi=bx; yhi.lo[1]=dom.hi[1]+1;
        ylo=box_grow(ylo,-1,0,0); yhi=box_grow(yhi,-1,0,0);
        FORBOX(ylo,i,j,k){
            if(bc.lo[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[1]*V(msku,i,j,0);
            else if(bc.lo[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dom.lo[1],k)*V(msku,i,j,0);
        }
        FORBOX(yhi,i,j,k){
            if(bc.hi[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[4]*V(msku,i,j,0);
            else if(bc.hi[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dom.hi[1],k)*V(msku,i,j,0);
        }
    }
    {                                                                                         /* :193-223 */
        Box zlo=bx; zlo.hi[2]=dom.lo[2]-1; Box zhi=bx; zhi.lo[2]=dom.hi[2]+1;
        FORBOX(zlo,i,j,k){ if(bc.lo[2]==BC_EXT_DIR) V(dest,i,j,k)=ext[2]*V(msku,i,j,0); else if(bc.lo[2]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,j,dom.lo[2])*V(msku,i,j,0); }
        FORBOX(zhi,i,j,k){ if(bc.hi[2]==BC_EXT_DIR) V(dest,i,j,k)=ext[5]*V(msku,i,j,0); else if(bc.hi[2]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,j,dom.hi[2])*V(msku,i,j,0); }
    }
    if((!perx||fx)&&(!pery||fx)){                                                             /* :225-265 */
        Box xlo=bx; xlo.hi[0]=dom.lo[0]; Box xhi=bx; xhi.lo[0]=dom.hi[0]+1;
        Box ylo=bx; ylo.hi[1]=dom.lo[1]-1; Box yhi=bx; yhi.lo[1]=dom.hi[1]+1;
        Box a=box_and(xlo,ylo), b=box_and(xlo,yhi), c=box_and(xhi,ylo), d=box_and(xhi,yhi);
        FORBOX(a,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dom.lo[1],k)+V(dest,dom.lo[0]+1,j,k));
        FORBOX(b,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dom.hi[1],k)+V(dest,dom.lo[0]+1,j,k));
        FORBOX(c,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dom.lo[1],k)+V(dest,dom.hi[0],j,k));
        FORBOX(d,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dom.hi[1],k)+V(dest,dom.hi[0],j,k));
    }
}

/* REMORA_BoundaryConditions_yvel.cpp:12-268 */
static void impose_yvel(const Geom *g,Fab dest,Box bx,Fab mskv,int bccomp)
{
    Box dom=g->domain; int perx=g->per[0], pery=g->per[1];
    BCRec bc=setbc(g->bcs[bccomp],bx,dom); const double *ext=g->extdir[bccomp];
    for(int d=0;d<3;d++){ unsupported(bc.lo[d]); unsupported(bc.hi[d]); }
    int fx=(bccomp==BCV_FOEXTRAP);
    if(!perx||fx){                                                                            /* :55-110 */
        Box xlo=bx; xlo.hi[0]=dom.lo[0]-1; Box xhi=bx; xhi.lo[0]=dom.hi[0]+1;
        xlo=box_grow(xlo,0,-1,0); xhi=box_grow(xhi,0,-1,0);
        FORBOX(xlo,i,j,k){
            if(bc.lo[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[0]*V(mskv,i,j,0);
            else if(bc.lo[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dom.lo[0],j,k)*V(mskv,i,j,0);
        }
        FORBOX(xhi,i,j,k){
            if(bc.hi[0]==BC_EXT_DIR) V(dest,i,j,k)=ext[3]*V(mskv,i,j,0);
            else if(bc.hi[0]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,dom.hi[0],j,k)*V(mskv,i,j,0);
        }
    }
    if(!pery||fx){                                                                            /* :112-192 */
        Box ylo=bx; ylo.hi[1]=dom.lo[1]-1; Box yhi=bx; yhi.lo[1]=dom.hi[1]+2;
        Box flo=bx; flo.lo[1]=flo.hi[1]=dom.lo[1]; Box fhi=bx; fhi.lo[1]=fhi.hi[1]=dom.hi[1]+1;
        ylo=box_grow(ylo,-1,0,0); yhi=box_grow(yhi,-1,0,0); flo=box_grow(flo,-1,0,0); fhi=box_grow(fhi,-1,0,0);
        FORBOX(flo,i,j,k){
            if(bc.lo[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[1]*V(mskv,i,j,0);
            else if(bc.lo[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dom.lo[1]+1,k)*V(mskv,i,j,0);
        }
        FORBOX(ylo,i,j,k){
            int inner=(bc.lo[1]==BC_FOEXTRAP)?1:0;
            if(bc.lo[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[1]*V(mskv,i,j,0);
            else if(bc.lo[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dom.lo[1]+inner,k)*V(mskv,i,j,0);
        }
        FORBOX(fhi,i,j,k){
            if(bc.hi[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[4]*V(mskv,i,j,0);
            else if(bc.hi[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dom.hi[1],k)*V(mskv,i,j,0);
        }
        FORBOX(yhi,i,j,k){
            int inner=(bc.hi[1]==BC_FOEXTRAP)?1:0;
            if(bc.hi[1]==BC_EXT_DIR) V(dest,i,j,k)=ext[4]*V(mskv,i,j,0);
            else if(bc.hi[1]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,dom.hi[1]+1-inner,k)*V(mskv,i,j,0);
        }
    }
    {                                                                                         /* :194-224 */
        Box zlo=bx; zlo.hi[2]=dom.lo[2]-1; Box zhi=bx; zhi.lo[2]=dom.hi[2]+1;
        FORBOX(zlo,i,j,k){ if(bc.lo[2]==BC_EXT_DIR) V(dest,i,j,k)=ext[2]*V(mskv,i,j,0); else if(bc.lo[2]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,j,dom.lo[2])*V(mskv,i,j,0); }
        FORBOX(zhi,i,j,k){ if(bc.hi[2]==BC_EXT_DIR) V(dest,i,j,k)=ext[5]*V(mskv,i,j,0); else if(bc.hi[2]==BC_FOEXTRAP) V(dest,i,j,k)=V(dest,i,j,dom.hi[2])*V(mskv,i,j,0); }
    }
    if((!perx||fx)&&(!pery||fx)){                                                             /* :226-265 */
        Box xlo=bx; xlo.hi[0]=dom.lo[0]-1; Box xhi=bx; xhi.lo[0]=dom.hi[0]+1;
        Box ylo=bx; ylo.hi[1]=dom.lo[1]; Box yhi=bx; yhi.lo[1]=dom.hi[1]+1;
        Box a=box_and(xlo,ylo), b=box_and(xlo,yhi), c=box_and(xhi,ylo), d=box_and(xhi,yhi);
        FORBOX(a,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dom.lo[1]+1,k)+V(dest,dom.lo[0],j,k));
        FORBOX(b,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dom.hi[1],k)+V(dest,dom.lo[0],j,k));
        FORBOX(c,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dom.lo[1]+1,k)+V(dest,dom.hi[0],j,k));
        FORBOX(d,i,j,k) V(dest,i,j,k)=0.5*(V(dest,i,dom.hi[1],k)+V(dest,dom.hi[0],j,k));
    }
}

/* REMORA_BoundaryConditions_zvel.cpp:13-124 */
static void impose_zvel(const Geom *g,Fab dest,Box bx,int bccomp)
{
    Box dom=g->domain; int perx=g->per[0], pery=g->per[1];
    BCRec bc=setbc(g->bcs[bccomp],bx,dom); const double *ext=g->extdir[bccomp];
    for(int d=0;d<3;d++){ unsupported(bc.lo[d]); unsupported(bc.hi[d]); }
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
void o_physbc(const Geom *g,MF *mf,Fab msk,int icomp,int ncomp,int bccomp,int n_not_fill)
{
    Box gdom=g->domain;
    for(int d=0;d<3;d++) if(mf->ix[d]) gdom.hi[d]+=1;
    /* 2-D arrays live on a k=0 slab of the domain */
    for(int d=0;d<2;d++) if(g->per[d] && bccomp!=BCV_FOEXTRAP){ gdom.lo[d]-=mf->ng[d]; gdom.hi[d]+=mf->ng[d]; }
    Box bx=fab_box(mf->f);
    if(box_contains_box(gdom,bx)) return;
    if(!mf->ix[0]&&!mf->ix[1]&&!mf->ix[2]) impose_cons(g,mf->f,bx,mf->valid,msk,icomp,ncomp,bccomp,n_not_fill);
    else if(mf->ix[0]) impose_xvel(g,fab_comp(mf->f,icomp),bx,msk,bccomp);
    else if(mf->ix[1]) impose_yvel(g,fab_comp(mf->f,icomp),bx,msk,bccomp);
    else if(mf->ix[2]) impose_zvel(g,fab_comp(mf->f,icomp),bx,bccomp);
}

void o_fillpatch(const Geom *g,MF *mf,Fab msk,int bccomp,int icomp,int ncomp)
{
    o_fill_boundary(g,mf,icomp,ncomp);
    o_physbc(g,mf,msk,icomp,ncomp,bccomp,0);
}
