// rh_ctx.hip -- context, BC table, barotropic filter weights, stretching curves, native state layout, timing.
#include "rh_common.h"
#include <cstdarg>
#include <cstdlib>

thread_local std::string g_rh_err;

int rh_fail(const char *fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_rh_err = buf;
    return 1;
}
extern "C" const char *rh_last_error(void) { return g_rh_err.c_str(); }
#define RH_BUILD_ID "r3_v8"
extern "C" const char *rh_build_id(void) { return RH_BUILD_ID; }

// ------------------------------------------------------------------------------------------------ timing
TScope::TScope(rh_ctx *ctx, const char *name) : c(ctx), on(ctx->timing)
{
    if (!on) return;
    auto it = c->tslot.find(name);
    if (it == c->tslot.end()) { slot = (int)c->tnames.size(); c->tslot[name] = slot; c->tnames.push_back(name); c->tev.emplace_back(); }
    else slot = it->second;
    auto get = [&](hipEvent_t &e) { if (!c->evpool.empty()) { e = c->evpool.back(); c->evpool.pop_back(); } else (void)hipEventCreate(&e); };
    get(a); get(b);
    (void)hipEventRecord(a, c->stream);
}
TScope::~TScope()
{
    if (!on) return;
    (void)hipEventRecord(b, c->stream);
    c->tev[slot].push_back({a, b});
}

extern "C" int rh_timing_enable(rh_ctx *c, int on) { c->timing = on != 0; return 0; }
extern "C" int rh_timing_reset(rh_ctx *c)
{
    RH_HIP(hipStreamSynchronize(c->stream));
    for (auto &v : c->tev) { for (auto &p : v) { c->evpool.push_back(p.first); c->evpool.push_back(p.second); } v.clear(); }
    return 0;
}
extern "C" int rh_timing_report(rh_ctx *c, char *buf, int buflen)
{
    RH_HIP(hipStreamSynchronize(c->stream));
    std::string s = "{";
    for (size_t q = 0; q < c->tnames.size(); q++) {
        double ms = 0; for (auto &p : c->tev[q]) { float t = 0; (void)hipEventElapsedTime(&t, p.first, p.second); ms += t; }
        char tmp[256]; snprintf(tmp, sizeof tmp, "%s\"%s\":{\"calls\":%zu,\"ms\":%.6f}", q ? "," : "", c->tnames[q].c_str(), c->tev[q].size(), ms);
        s += tmp;
    }
    s += "}";
    if ((int)s.size() + 1 > buflen) return rh_fail("rh_timing_report: buffer too small (%zu needed)", s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}

// ------------------------------------------------------------------------------------------------ small device utilities
__global__ void k_setval(double *p, long n, double v)
{
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x; const long st = (long)gridDim.x * blockDim.x;
    for (; q < n; q += st) p[q] = v;
}
__global__ void k_copy(double *d, const double *s, long n)
{
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x; const long st = (long)gridDim.x * blockDim.x;
    for (; q < n; q += st) d[q] = s[q];
}
static inline int nblk(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

int dev_setval(rh_ctx *c, const rh_array4 &a, int icomp, int ncomp, double v)
{
    long per = a4_size(a) / a.ncomp; long n = per * ncomp; if (n <= 0) return 0;
    k_setval<<<nblk(n), 256, 0, c->stream>>>(a.p + per * icomp, n, v);
    RH_LAUNCH_CHECK(); return 0;
}
int dev_copy(rh_ctx *c, const rh_array4 &dst, const rh_array4 &src)
{
    if (a4_size(dst) != a4_size(src)) return rh_fail("dev_copy: size mismatch");
    long n = a4_size(dst); if (n <= 0) return 0;
    k_copy<<<nblk(n), 256, 0, c->stream>>>(dst.p, src.p, n);
    RH_LAUNCH_CHECK(); return 0;
}

MFV mfv(const rh_ctx *c, const rh_array4 &a, int ix0, int ix1, int ix2)
{
    MFV m; m.a = a; m.ix[0] = ix0; m.ix[1] = ix1; m.ix[2] = ix2;
    m.valid = c->bx;
    // 2-D arrays live on the k = 0 slab; k-limited coefficient arrays keep their own k range
    m.valid.lo[2] = a.lo[2]; m.valid.hi[2] = a.hi[2];
    if (ix0) m.valid.hi[0] += 1;
    if (ix1) m.valid.hi[1] += 1;
    for (int d = 0; d < 2; d++) m.ng[d] = m.valid.lo[d] - a.lo[d];
    m.ng[2] = 0;
    return m;
}

// ------------------------------------------------------------------------------------------------ set_weights
// REMORA::set_weights, Source/TimeIntegration/REMORA_set_weights.cpp:10-171 (power-law shape filter, host scalar code)
static int set_weights(int ndtfast, std::vector<double> &weight1, std::vector<double> &weight2)
{
    double gamma, scale, wsum, shift, cff;
    const double Falpha = 2.0, Fbeta = 4.0, Fgamma = 0.284;
    int nfast = 0;
    weight1.assign(2 * ndtfast + 1, 0.0); weight2.assign(2 * ndtfast + 1, 0.0);
    scale = (Falpha + 1.0) * (Falpha + Fbeta + 1.0) / ((Falpha + 2.0) * (Falpha + Fbeta + 2.0) * (double)ndtfast);
    gamma = Fgamma * std::max(0.0, 1.0 - 10.0 / (double)ndtfast);
    for (int iter = 1; iter <= 16; iter++) {
        nfast = 0;
        for (int i = 1; i <= 2 * ndtfast; i++) {
            cff = scale * (double)i;
            weight1[i-1] = (double)(pow(cff, Falpha) - pow(cff, (Falpha + Fbeta))) - gamma * cff;
            if (weight1[i-1] > 0.0) nfast = i;
            if ((nfast > 0) && (weight1[i-1] < 0.0)) weight1[i-1] = 0.0;
        }
        wsum = 0.0; shift = 0.0;
        for (int i = 1; i <= nfast; i++) { wsum = wsum + weight1[i-1]; shift = shift + weight1[i-1] * (double)i; }
        scale *= shift / (wsum * (double)ndtfast);
    }
    for (int iter = 1; iter <= ndtfast; iter++) {
        wsum = 0.0; shift = 0.0;
        for (int i = 1; i <= nfast; i++) { wsum = wsum + weight1[i-1]; shift = shift + (double)i * weight1[i-1]; }
        shift = shift / wsum;
        cff = (double)ndtfast - shift;
        if (cff > 1.0) {
            nfast = nfast + 1;
            for (int i = nfast; i >= 2; i--) weight1[i-1] = weight1[i-2];
            weight1[0] = 0.0;
        } else if (cff > 0.0) {
            wsum = 1.0 - cff;
            for (int i = nfast; i >= 2; i--) weight1[i-1] = wsum * weight1[i-1] + cff * weight1[i-2];
            weight1[0] = wsum * weight1[0];
        } else if (cff < -1.0) {
            nfast = nfast - 1;
            for (int i = 1; i <= nfast; i++) weight1[i-1] = weight1[i];
            weight1[nfast] = 0.0;
        } else if (cff < 0.0) {
            wsum = 1.0 + cff;
            for (int i = 1; i <= nfast - 1; i++) weight1[i-1] = wsum * weight1[i-1] - cff * weight1[i];
            weight1[nfast-1] = wsum * weight1[nfast-1];
        }
    }
    for (int j = 1; j <= nfast; j++) { cff = weight1[j-1]; for (int i = 1; i <= j; i++) weight2[i-1] = weight2[i-1] + cff; }
    wsum = 0.0; cff = 0.0;
    for (int i = 1; i <= nfast; i++) { wsum = wsum + weight1[i-1]; cff = cff + weight2[i-1]; }
    wsum = 1.0 / wsum; cff = 1.0 / cff;
    for (int i = 1; i <= nfast; i++) { weight1[i-1] = wsum * weight1[i-1]; weight2[i-1] = cff * weight2[i-1]; }
    return nfast;
}

// k-only part of REMORA::stretch_transform, Source/Utils/REMORA_DepthStretchTransform.H:81-139 (N = nz there).
// The reference evaluates cosh/exp per (i,j,k); the curves depend on k alone, so they are tabulated once.
static void stretch_coeffs(int k, int N, double theta_s, double theta_b, double *sc_w_o, double *Cs_w_o, double *sc_r_o, double *Cs_r_o)
{
    const double ds = 1.0 / (double)N;
    double Csur, Cbot, sc_r, sc_w, Cs_r, Cs_w;
    if (k == N) { sc_w = 0.0; Cs_w = 0.0; }
    else if (k == 0) { sc_w = -1.0; Cs_w = -1.0; }
    else {
        sc_w = ds * (k - N);
        if (theta_s > 0.0) Csur = (1.0 - cosh(theta_s * sc_w)) / (cosh(theta_s) - 1.0);
        else Csur = -sc_w * sc_w;
        if (theta_b > 0.0) { Cbot = (exp(theta_b * Csur) - 1.0) / (1.0 - exp(-theta_b)); Cs_w = Cbot; }
        else Cs_w = Csur;
    }
    sc_r = ds * (k - N + 0.5);
    if (theta_s > 0.0) Csur = (1.0 - cosh(theta_s * sc_r)) / (cosh(theta_s) - 1.0);
    else Csur = -sc_r * sc_r;
    if (theta_b > 0.0) { Cbot = (exp(theta_b * Csur) - 1.0) / (1.0 - exp(-theta_b)); Cs_r = Cbot; }
    else Cs_r = Csur;
    *sc_w_o = sc_w; *Cs_w_o = Cs_w; *sc_r_o = sc_r; *Cs_r_o = Cs_r;
}

// REMORA::init_bcs, Source/Initialization/REMORA_init_bcs.cpp:201-679: phys_bc_type -> domain_bcs_type (the mathematical type of every
// BCVars component on every face) and m_bc_extdir_vals
static int translate_bcs(rh_ctx *c)
{
    const int BOGUS = -666;
    auto put = [&](int v, int dir, int hi, int t) { if (hi) c->bcs[v].hi[dir] = t; else c->bcs[v].lo[dir] = t; };
    for (int v = 0; v < RH_BCV_NUM; v++) for (int d = 0; d < 3; d++) { c->bcs[v].lo[d] = BOGUS; c->bcs[v].hi[d] = BOGUS; }
    for (int v = 0; v < RH_BCV_NUM; v++) for (int o = 0; o < 6; o++) c->extdir[v][o] = 0.0;
    for (int o = 0; o < 6; o++) { c->extdir[0][o] = 1.e19; c->extdir[1][o] = 1.e20; c->extdir[2][o] = 1.e21; }   // :161-163
    for (int v = 0; v < 10; v++) for (int o = 0; o < 6; o++) {
        const int dir = o % 3, kind = c->phys[v][o];
        const bool vel = (v == RH_BCV_XVEL || v == RH_BCV_YVEL || v == RH_BCV_ZVEL);
        if (kind == RH_PBC_INFLOW && (vel || v == RH_BCV_CONS + 2)) c->extdir[v][o] = c->physval[v][o];                       // :27-36
        if (kind == RH_PBC_NO_SLIP_WALL && vel) c->extdir[v][o] = (v - RH_BCV_XVEL == dir) ? 0.0 : c->physval[v][o];           // :43-54
    }
    for (int o = 0; o < 6; o++) {                                                                // :210-318 velocities
        const int dir = o % 3, hi = o / 3;
        for (int i = 0; i < 2; i++) {
            const int v = RH_BCV_XVEL + i;
            switch (c->phys[v][o]) {
            case RH_PBC_SYMMETRY:     put(v, dir, hi, RH_BC_REFLECT_EVEN); if (i == 1) put(RH_BCV_XVEL + dir, dir, hi, RH_BC_REFLECT_ODD); break;
            case RH_PBC_OUTFLOW:      put(v, dir, hi, RH_BC_FOEXTRAP); break;
            case RH_PBC_INFLOW:
            case RH_PBC_NO_SLIP_WALL: put(v, dir, hi, RH_BC_EXT_DIR); break;
            case RH_PBC_SLIP_WALL:    put(v, dir, hi, RH_BC_FOEXTRAP); if (i == 1) put(RH_BCV_XVEL + dir, dir, hi, RH_BC_EXT_DIR); break;   // only the normal component is ext_dir
            case RH_PBC_PERIODIC:     put(v, dir, hi, RH_BC_INT_DIR); break;
            case RH_PBC_CLAMPED:      put(v, dir, hi, RH_BC_CLAMPED); break;
            case RH_PBC_ORLANSKI_RAD: put(v, dir, hi, RH_BC_ORLANSKI_RAD); break;
            case RH_PBC_ORLANSKI_RAD_NUDGE: put(v, dir, hi, RH_BC_ORLANSKI_RAD_NUDGE); break;
            default: return rh_fail("init_bcs: Velocity boundary condition not validly specified (kind %d, face %d)", c->phys[v][o], o);
            }
        }
        put(RH_BCV_ZVEL, dir, hi, RH_BC_FOEXTRAP);                                               // :311-316
    }
    for (int o = 0; o < 6; o++) {                                                                // :326-411 tracers
        const int dir = o % 3, hi = o / 3;
        for (int i = 0; i < RH_NCONS; i++) {
            int t;
            switch (c->phys[RH_BCV_CONS + i][o]) {
            case RH_PBC_SYMMETRY: t = RH_BC_REFLECT_EVEN; break;
            case RH_PBC_OUTFLOW: case RH_PBC_NO_SLIP_WALL: case RH_PBC_SLIP_WALL: t = RH_BC_FOEXTRAP; break;
            case RH_PBC_INFLOW: t = RH_BC_EXT_DIR; break;
            case RH_PBC_PERIODIC: t = RH_BC_INT_DIR; break;
            case RH_PBC_CLAMPED: t = RH_BC_CLAMPED; break;
            case RH_PBC_ORLANSKI_RAD: t = RH_BC_ORLANSKI_RAD; break;
            case RH_PBC_ORLANSKI_RAD_NUDGE: t = RH_BC_ORLANSKI_RAD_NUDGE; break;
            default: return rh_fail("init_bcs: Scalar/tracer boundary condition not validly specified (kind %d, face %d)", c->phys[RH_BCV_CONS + i][o], o);
            }
            put(RH_BCV_CONS + i, dir, hi, t);
        }
    }
    for (int o = 0; o < 6; o++) {                                                                // :420-543 ubar, vbar and their simplified 2-D twins
        const int dir = o % 3, hi = o / 3;
        for (int i = 0; i < 2; i++) {
            const int v = RH_BCV_UBAR + i, v2 = RH_BCV_U2D_SIMPLE + i;
            const bool nrm = (i == 1 && dir != 2);
            switch (c->phys[v][o]) {
            case RH_PBC_SYMMETRY:
                put(v, dir, hi, RH_BC_REFLECT_EVEN); put(v2, dir, hi, RH_BC_REFLECT_EVEN);
                if (nrm) { put(RH_BCV_UBAR + dir, dir, hi, RH_BC_REFLECT_ODD); put(RH_BCV_U2D_SIMPLE + dir, dir, hi, RH_BC_REFLECT_ODD); }
                break;
            case RH_PBC_OUTFLOW:      put(v, dir, hi, RH_BC_FOEXTRAP); put(v2, dir, hi, RH_BC_FOEXTRAP); break;
            case RH_PBC_INFLOW:
            case RH_PBC_NO_SLIP_WALL: put(v, dir, hi, RH_BC_EXT_DIR); put(v2, dir, hi, RH_BC_EXT_DIR); break;
            case RH_PBC_SLIP_WALL:
                put(v, dir, hi, RH_BC_FOEXTRAP); put(v2, dir, hi, RH_BC_FOEXTRAP);
                if (nrm) { put(RH_BCV_UBAR + dir, dir, hi, RH_BC_EXT_DIR); put(RH_BCV_U2D_SIMPLE + dir, dir, hi, RH_BC_EXT_DIR); }
                break;
            case RH_PBC_PERIODIC:     put(v, dir, hi, RH_BC_INT_DIR); put(v2, dir, hi, RH_BC_INT_DIR); break;
            case RH_PBC_CLAMPED:      put(v, dir, hi, RH_BC_CLAMPED); put(v2, dir, hi, RH_BC_FOEXTRAP); break;
            case RH_PBC_FLATHER:
                put(v, dir, hi, RH_BC_CHAPMAN); put(v2, dir, hi, RH_BC_FOEXTRAP);
                if (nrm) { put(RH_BCV_UBAR + dir, dir, hi, RH_BC_FLATHER); put(RH_BCV_U2D_SIMPLE + dir, dir, hi, RH_BC_FOEXTRAP); }
                break;
            default: return rh_fail("init_bcs: ubar or vbar boundary condition not validly specified (kind %d, face %d)", c->phys[v][o], o);
            }
        }
    }
    for (int o = 0; o < 6; o++) {                                                                // :551-628 zeta, tke
        const int dir = o % 3, hi = o / 3;
        for (int i = 0; i < 2; i++) {
            int t;
            switch (c->phys[RH_BCV_ZETA + i][o]) {
            case RH_PBC_SYMMETRY: t = RH_BC_REFLECT_EVEN; break;
            case RH_PBC_OUTFLOW: case RH_PBC_NO_SLIP_WALL: case RH_PBC_SLIP_WALL: t = RH_BC_FOEXTRAP; break;
            case RH_PBC_INFLOW: t = RH_BC_EXT_DIR; break;
            case RH_PBC_PERIODIC: t = RH_BC_INT_DIR; break;
            case RH_PBC_CHAPMAN: t = RH_BC_CHAPMAN; break;
            case RH_PBC_CLAMPED: t = RH_BC_CLAMPED; break;
            default: return rh_fail("init_bcs: Free surface (zeta) boundary condition not validly specified (kind %d, face %d)", c->phys[RH_BCV_ZETA + i][o], o);
            }
            put(RH_BCV_ZETA + i, dir, hi, t);
        }
    }
    for (int o = 0; o < 6; o++) {                                                                // :636-667
        put(RH_BCV_FOEXTRAP_PERIODIC, o % 3, o / 3, RH_BC_FOEXTRAP);
        put(RH_BCV_FOEXTRAP, o % 3, o / 3, RH_BC_FOEXTRAP);
    }
    return 0;
}
// the state REMORA::init_bcs reaches for the pinned inputs: periodic where the geometry is, SlipWall elsewhere (z: always, :130)
static int init_bcs(rh_ctx *c)
{
    for (int v = 0; v < 10; v++) for (int o = 0; o < 6; o++) {
        const int dir = o % 3;
        c->phys[v][o] = (dir < 2 && c->prm.periodic[dir]) ? RH_PBC_PERIODIC : RH_PBC_SLIP_WALL;
        c->physval[v][o] = 0.0;
    }
    return translate_bcs(c);
}
extern "C" int rh_set_bc(rh_ctx *c, int bcvar, int face, int kind, double value)
{
    if (!c) return rh_fail("rh_set_bc: null context");
    if (!(face == 0 || face == 1 || face == 3 || face == 4)) return rh_fail("rh_set_bc: face %d is not one of 0 xlo, 1 ylo, 3 xhi, 4 yhi", face);
    if (bcvar < -1 || bcvar >= 10) return rh_fail("rh_set_bc: variable %d is not one of the ten user variables", bcvar);
    if (kind < 0 || kind >= RH_PBC_UNDEFINED) return rh_fail("rh_set_bc: This BC type is unknown (%d)", kind);                    // :117-121
    const int dir = face % 3;
    if (c->prm.periodic[dir] && kind != RH_PBC_PERIODIC) return rh_fail("rh_set_bc: Wrong BC type for periodic boundary");         // :107-115
    if (!c->prm.periodic[dir] && kind == RH_PBC_PERIODIC) return rh_fail("rh_set_bc: Periodic boundary specified in a non-periodic direction");   // :96-98
    int saved[10][6]; double savedv[10][6];
    memcpy(saved, c->phys, sizeof saved); memcpy(savedv, c->physval, sizeof savedv);
    for (int v = (bcvar < 0 ? 0 : bcvar); v <= (bcvar < 0 ? 9 : bcvar); v++) {
        if (kind == RH_PBC_CHAPMAN && v != RH_BCV_ZETA) { memcpy(c->phys, saved, sizeof saved); memcpy(c->physval, savedv, sizeof savedv); return rh_fail("rh_set_bc: Chapman BC can only be applied to zeta"); }   // :71-73
        if (kind == RH_PBC_FLATHER && !(v == RH_BCV_UBAR || v == RH_BCV_VBAR)) { memcpy(c->phys, saved, sizeof saved); memcpy(c->physval, savedv, sizeof savedv); return rh_fail("rh_set_bc: Flather BC can only be applied to ubar or vbar"); }   // :80-82
        c->phys[v][face] = kind; c->physval[v][face] = value;
    }
    if (translate_bcs(c)) { memcpy(c->phys, saved, sizeof saved); memcpy(c->physval, savedv, sizeof savedv); (void)translate_bcs(c); return 1; }
    return 0;
}
extern "C" int rh_get_bc(const rh_ctx *c, int bccomp, int out6[6])
{
    if (!c || bccomp < 0 || bccomp >= RH_BCV_NUM) return rh_fail("rh_get_bc: component %d", bccomp);
    for (int d = 0; d < 3; d++) { out6[d] = c->bcs[bccomp].lo[d]; out6[3 + d] = c->bcs[bccomp].hi[d]; }
    return 0;
}

// ------------------------------------------------------------------------------------------------ native state layout
// Shapes and ghost widths: Source/Initialization/REMORA_make_new_level.cpp:259-282,381-389,426-503 (NGROW = 2).
// kind: 0 = 3-D cell/face array over the full column, 1 = 2-D (k = 0 slab), 2 = k-limited coefficient array (k = 0..1:
// visc2_p / visc2_r / diff2 are 3-D in the reference but only ever indexed at k = 0, and at k = 1 through the
// 3-index access of REMORA_t3dmix.cpp:34,43)
struct FieldSpec { rh_array4 rh_state::*m; int ix0, ix1, ix2, nc, ng0, ng1, kind; int gls_only = 0; };
static const FieldSpec g_fields[] = {
    {&rh_state::cons_old,0,0,0,RH_NCONS,3,3,0}, {&rh_state::cons_new,0,0,0,RH_NCONS,3,3,0},
    {&rh_state::xvel_old,1,0,0,1,3,3,0}, {&rh_state::xvel_new,1,0,0,1,3,3,0},
    {&rh_state::yvel_old,0,1,0,1,3,3,0}, {&rh_state::yvel_new,0,1,0,1,3,3,0},
    {&rh_state::zvel_old,0,0,1,1,3,3,0}, {&rh_state::zvel_new,0,0,1,1,3,3,0},
    {&rh_state::Zt_avg1,0,0,0,1,3,3,1}, {&rh_state::h,0,0,0,2,3,3,1},
    {&rh_state::ubar,1,0,0,3,2,2,1}, {&rh_state::vbar,0,1,0,3,2,2,1},
    {&rh_state::ru,1,0,0,2,2,2,0}, {&rh_state::rv,0,1,0,2,2,2,0},
    {&rh_state::ru2d,1,0,0,2,2,2,1}, {&rh_state::rv2d,0,1,0,2,2,2,1},
    {&rh_state::mskr,0,0,0,1,3,3,1}, {&rh_state::msku,1,0,0,1,3,3,1}, {&rh_state::mskv,0,1,0,1,3,3,1}, {&rh_state::mskp,1,1,0,1,3,3,1},
    {&rh_state::z_w,0,0,1,1,3,3,0}, {&rh_state::z_r,0,0,0,1,3,3,0}, {&rh_state::Hz,0,0,0,1,3,3,0},
    {&rh_state::Huon,1,0,0,1,2,2,0}, {&rh_state::Hvom,0,1,0,1,2,2,0},
    {&rh_state::Akv,0,0,1,1,2,2,0}, {&rh_state::Akt,0,0,1,RH_NCONS,2,2,0},
    {&rh_state::visc2_p,0,0,0,1,2,2,2}, {&rh_state::visc2_r,0,0,0,1,2,2,2}, {&rh_state::diff2,0,0,0,RH_NCONS,2,2,2},
    {&rh_state::rufrc,1,0,0,2,2,2,1}, {&rh_state::rvfrc,0,1,0,2,2,2,1},
    {&rh_state::sustr,1,0,0,1,2,2,1}, {&rh_state::svstr,0,1,0,1,2,2,1}, {&rh_state::rdrag,0,0,0,1,2,2,1},
    {&rh_state::bustr,1,0,0,1,2,2,1}, {&rh_state::bvstr,0,1,0,1,2,2,1},
    {&rh_state::DU_avg1,1,0,0,1,2,2,1}, {&rh_state::DU_avg2,1,0,0,1,2,2,1},
    {&rh_state::DV_avg1,0,1,0,1,2,2,1}, {&rh_state::DV_avg2,0,1,0,1,2,2,1},
    {&rh_state::rubar,1,0,0,4,2,2,1}, {&rh_state::rvbar,0,1,0,4,2,2,1}, {&rh_state::rzeta,0,0,0,4,2,2,1},
    {&rh_state::zeta,0,0,0,3,3,3,1},
    {&rh_state::pm,0,0,0,1,3,4,1}, {&rh_state::pn,0,0,0,1,4,3,1}, {&rh_state::fcor,0,0,0,1,3,3,1},
    {&rh_state::sstore,0,0,0,RH_NCONS,2,2,0},
    {&rh_state::rhoS,0,0,0,1,2,2,1}, {&rh_state::rhoA,0,0,0,1,2,2,1}, {&rh_state::bvf,0,0,1,1,2,2,0},
    // GLS closure, make_new_level.cpp:499-503
    {&rh_state::tke,0,0,1,3,2,2,0,1}, {&rh_state::gls,0,0,1,3,2,2,0,1}, {&rh_state::Lscale,0,0,1,1,2,2,0,1},
    {&rh_state::Akk,0,0,1,1,2,2,0,1}, {&rh_state::Akp,0,0,1,1,2,2,0,1},
};

static rh_array4 shape_of(const rh_ctx *c, int ix0, int ix1, int ix2, int nc, int ng0, int ng1, int kind)
{
    rh_array4 a; a.p = nullptr; a.ncomp = nc;
    rh_box b = c->bx;
    if (ix0) b.hi[0] += 1;
    if (ix1) b.hi[1] += 1;
    if (ix2) b.hi[2] += 1;
    if (kind == 1) { b.lo[2] = 0; b.hi[2] = 0; }
    if (kind == 2) { b.lo[2] = 0; b.hi[2] = 1; }
    b = bx_grow(b, ng0, ng1, 0);
    for (int d = 0; d < 3; d++) { a.lo[d] = b.lo[d]; a.hi[d] = b.hi[d]; }
    return a;
}
static inline long align32(long ndbl) { return (ndbl + 31) / 32 * 32; }    // 256-B aligned carve

extern "C" long rh_state_bytes(const rh_ctx *c)
{
    long n = 0;
    for (const FieldSpec &f : g_fields) {
        if (f.gls_only && c->prm.vmix_mode != RH_VMIX_GLS) continue;
        n += align32(a4_size(shape_of(c, f.ix0, f.ix1, f.ix2, f.nc, f.ng0, f.ng1, f.kind)));
    }
    return n * 8;
}
extern "C" int rh_state_bind(const rh_ctx *c, void *device_base, rh_state *out)
{
    if (((size_t)device_base) & 255) return rh_fail("rh_state_bind: base must be 256-byte aligned");
    const_cast<rh_ctx *>(c)->vmix_akv_done = const_cast<rh_ctx *>(c)->vmix_akt_done = nullptr;     // whatever the block holds, it is not yet the constants
    double *p = (double *)device_base;
    for (const FieldSpec &f : g_fields) {
        if (f.gls_only && c->prm.vmix_mode != RH_VMIX_GLS) { rh_array4 z; memset(&z, 0, sizeof z); z.hi[0] = z.hi[1] = z.hi[2] = -1; out->*(f.m) = z; continue; }
        rh_array4 a = shape_of(c, f.ix0, f.ix1, f.ix2, f.nc, f.ng0, f.ng1, f.kind);
        a.p = p; p += align32(a4_size(a));
        out->*(f.m) = a;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ create / destroy
extern "C" int rh_create(const rh_params *prm, void *stream, rh_ctx **out)
{
    if (!prm || !out) return rh_fail("rh_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return rh_fail("rh_create: no HIP device visible -- this library has no CPU path");
    rh_ctx *c = new rh_ctx();
    c->prm = *prm; c->stream = (hipStream_t)stream;
    for (int d = 0; d < 3; d++) { c->dom.lo[d] = prm->domain_lo[d]; c->dom.hi[d] = prm->domain_hi[d]; c->bx.lo[d] = prm->box_lo[d]; c->bx.hi[d] = prm->box_hi[d]; }
    c->nz = c->dom.hi[2] - c->dom.lo[2] + 1; c->N = c->nz - 1;
    // every failure leaves through here: device memory allocated so far is freed before the context
    auto bad = [&](const char *m) {
        const std::string msg = m;                          // m may point into g_rh_err
        if (c->d_sc) (void)hipFree(c->d_sc);
        if (c->d_vmixrow) (void)hipFree(c->d_vmixrow);
        if (c->d_work) (void)hipFree(c->d_work);
        delete c;
        return rh_fail("rh_create: %s", msg.c_str());
    };
    auto bad_hip = [&](const char *what, hipError_t e) { const std::string m = std::string(what) + " failed: " + hipGetErrorString(e); return bad(m.c_str()); };
    if (c->dom.lo[2] != 0) return bad("domain must start at k = 0");
    if (c->nz < 3) return bad("at least 3 vertical levels are required (4th-order vertical fluxes index k = N-2)");
    if (c->bx.lo[2] != c->dom.lo[2] || c->bx.hi[2] != c->dom.hi[2]) return bad("boxes must span the full vertical (main.cpp:21-23)");
    if (!bx_contains(c->dom, c->bx)) return bad("box outside domain");
    if (prm->ndtfast < 1) return bad("ndtfast < 1");
    // i-j decomposition: a tensor-product process grid; every rank owns one box, contiguous in i and j
    for (int d = 0; d < 2; d++) {
        const int P = prm->nranks[d] < 1 ? 1 : prm->nranks[d], r = prm->rank_xy[d];
        if (r < 0 || r >= P) return bad("rank_xy outside the process grid");
        if (P == 1 && (c->bx.lo[d] != c->dom.lo[d] || c->bx.hi[d] != c->dom.hi[d])) return bad("a direction with a single rank must be owned whole");
        if (P > 1) {
            if ((r == 0) != (c->bx.lo[d] == c->dom.lo[d]) || (r == P - 1) != (c->bx.hi[d] == c->dom.hi[d])) return bad("box_lo/box_hi inconsistent with rank_xy");
            if (c->bx.hi[d] - c->bx.lo[d] + 1 < 4) return bad("a box must be at least 4 cells wide in a decomposed direction (ghost width 3 + shared face)");
        }
    }
    for (int d = 0; d < 2; d++) if (prm->periodic[d] && (c->dom.hi[d] - c->dom.lo[d] + 1) < 4) return bad("periodic direction shorter than the ghost width");
    c->bcq = bc_queue_new(); c->haloq = halo_queue_new();
    c->nfast = set_weights(prm->ndtfast, c->w1, c->w2);
    if (init_bcs(c)) return bad(rh_last_error());
    c->unfused = getenv("RH_UNFUSED") && atoi(getenv("RH_UNFUSED")) != 0;
    c->noregsolve = getenv("RH_NOREGSOLVE") && atoi(getenv("RH_NOREGSOLVE")) != 0;
    c->nofuse2d = getenv("RH_NOFUSE2D") && atoi(getenv("RH_NOFUSE2D")) != 0;
    c->gls_libm_pow = getenv("RH_GLS_LIBM_POW") && atoi(getenv("RH_GLS_LIBM_POW")) != 0;
    c->no_async = getenv("RH_NO_ASYNC") && atoi(getenv("RH_NO_ASYNC")) != 0;
    c->async_rccl = getenv("RH_ASYNC_RCCL") && atoi(getenv("RH_ASYNC_RCCL")) != 0;
    c->no_mffuse = getenv("RH_NO_MFFUSE") && atoi(getenv("RH_NO_MFFUSE")) != 0;
    c->no_split2d = getenv("RH_NO_SPLIT2D") && atoi(getenv("RH_NO_SPLIT2D")) != 0;
    c->no_bc2d = getenv("RH_NO_BC2D") && atoi(getenv("RH_NO_BC2D")) != 0;
    c->poison = getenv("RH_POISON") && atoi(getenv("RH_POISON")) != 0;
    c->no_rhofuse = getenv("RH_NO_RHOFUSE") && atoi(getenv("RH_NO_RHOFUSE")) != 0;
    c->vmix_rewrite = getenv("RH_VMIX_REWRITE") && atoi(getenv("RH_VMIX_REWRITE")) != 0;
    c->no_ckonst = getenv("RH_NO_CKONST") && atoi(getenv("RH_NO_CKONST")) != 0;
    c->no_k2mode = getenv("RH_NO_K2MODE") && atoi(getenv("RH_NO_K2MODE")) != 0;   // k2d_fused always reads its time-stepping modes from the arguments (A/B parity and timing)
    c->no_xshare = getenv("RH_NO_XSHARE") && atoi(getenv("RH_NO_XSHARE")) != 0;   // k_rhs_uv_tiled evaluates all eight horizontal fluxes per thread (A/B parity and timing)
    c->no_flatstretch = getenv("RH_NO_FLATSTRETCH") && atoi(getenv("RH_NO_FLATSTRETCH")) != 0;   // stretch_transform always as the tiled kernel (A/B parity and timing)

    // stretching curves (here N = nz, DepthStretchTransform.H:49)
    const int nz = c->nz;
    c->h_sc.assign(4 * (nz + 1), 0.0);
    for (int k = 0; k <= nz; k++)
        stretch_coeffs(k, nz, prm->theta_s, prm->theta_b, &c->h_sc[0 * (nz + 1) + k], &c->h_sc[1 * (nz + 1) + k], &c->h_sc[2 * (nz + 1) + k], &c->h_sc[3 * (nz + 1) + k]);
    hipError_t e = hipMalloc((void **)&c->d_sc, sizeof(double) * c->h_sc.size());
    if (e != hipSuccess) { c->d_sc = nullptr; return bad_hip("hipMalloc", e); }
    e = hipMemcpy(c->d_sc, c->h_sc.data(), sizeof(double) * c->h_sc.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bad_hip("hipMemcpy", e);

    if (prm->vmix_mode == RH_VMIX_CONST) {                   // the four constant rows of rh_ctx::d_vmixrow, as long as a row of Akv
        const rh_array4 ak = shape_of(c, 0, 0, 1, 1, 2, 2, 0);
        c->vmixrow_len = (int)align32(ak.hi[0] - ak.lo[0] + 1 + 8);
        std::vector<double> rows((size_t)4 * c->vmixrow_len);
        for (int w = 0; w < 4; w++)
            for (int q = 0; q < c->vmixrow_len; q++) rows[(size_t)w * c->vmixrow_len + q] = w == 0 ? prm->Akv_const : prm->Akt_const[w - 1];
        e = hipMalloc((void **)&c->d_vmixrow, sizeof(double) * rows.size());
        if (e != hipSuccess) { c->d_vmixrow = nullptr; return bad_hip("hipMalloc", e); }
        e = hipMemcpy(c->d_vmixrow, rows.data(), sizeof(double) * rows.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) return bad_hip("hipMemcpy", e);
    }
    // work space: rho (setup_step.cpp:63), P (prsgrd.cpp:39), W (setup_step.cpp:210, advance_3d.cpp:230),
    // CF/DC of the tridiagonal solve (advance_3d.cpp:49-51, vert_visc_3d.cpp), DUon/DVom (advance_2d.cpp:105-106)
    c->rho  = shape_of(c, 0, 0, 0, 1, 2, 2, 0);
    c->P    = shape_of(c, 0, 0, 0, 1, 1, 1, 0);
    c->W    = shape_of(c, 0, 0, 1, 1, 3, 3, 0);
    c->wCF  = shape_of(c, 1, 1, 1, 1, 0, 0, 0);
    c->wDC  = shape_of(c, 1, 1, 1, 1, 0, 0, 0);
    c->DUon = shape_of(c, 1, 0, 0, 1, 2, 2, 1);
    c->DVom = shape_of(c, 0, 1, 0, 1, 2, 2, 1);
    // 2-D intermediates of one advance_2d call (advance_2d.cpp:233-248): Drhs on grow 3, the rest on grow 2
    c->Drhs = shape_of(c, 0, 0, 0, 1, 3, 3, 1);
    c->Dnew = shape_of(c, 0, 0, 0, 1, 2, 2, 1); c->zwrk = c->Dnew; c->gzeta = c->Dnew; c->gzeta2 = c->Dnew; c->gzetaSA = c->Dnew;
    std::vector<rh_array4 *> ws = {&c->rho, &c->P, &c->W, &c->wCF, &c->wDC, &c->DUon, &c->DVom, &c->Drhs, &c->Dnew, &c->zwrk, &c->gzeta, &c->gzeta2, &c->gzetaSA};
    if (prm->vmix_mode == RH_VMIX_GLS) {
        // gls_corrector's scratch (REMORA_gls.cpp:355-376): three w-point arrays on the grow-1 box, the cached shear, buoy2, shear2
        c->gW1 = shape_of(c, 0, 0, 1, 1, 1, 1, 0); c->gW2 = c->gW1; c->gW3 = c->gW1;
        c->gS2C = shape_of(c, 0, 0, 0, 1, 2, 2, 0);
        c->gB2 = shape_of(c, 0, 0, 0, 1, 0, 0, 0); c->gS2 = c->gB2;
        for (rh_array4 *a : {&c->gW1, &c->gW2, &c->gW3, &c->gS2C, &c->gB2, &c->gS2}) ws.push_back(a);
        const rh_gls_params &G = prm->gls;
        if (G.stability < RH_GLS_CANUTO_A || G.stability > RH_GLS_GALPERIN) return bad("unknown gls stability type");
        if (G.gls_n == 0.0 || !(G.gls_cmu0 > 0.0) || !(G.gls_sigk > 0.0) || !(G.gls_sigp > 0.0) || !(G.gls_Kmin > 0.0) || !(G.gls_Pmin > 0.0))
            return bad("gls parameters: gls_n must be non-zero; cmu0, sigk, sigp, Kmin, Pmin positive");
    }
    long n = 0; for (rh_array4 *a : ws) n += align32(a4_size(*a));
    c->work_bytes = n * 8;
    e = hipMalloc((void **)&c->d_work, c->work_bytes);
    if (e != hipSuccess) { c->d_work = nullptr; char m[96]; snprintf(m, sizeof m, "hipMalloc(%ld bytes of work space)", n * 8); return bad_hip(m, e); }
    double *p = c->d_work; for (rh_array4 *a : ws) { a->p = p; p += align32(a4_size(*a)); }
    e = hipMemsetAsync(c->d_work, 0, c->work_bytes, c->stream);
    if (e != hipSuccess) return bad_hip("hipMemsetAsync", e);
    if (diag_scratch_init(c)) { const std::string m = g_rh_err; return bad(m.c_str()); }
    *out = c;
    return 0;
}

extern "C" int rh_destroy(rh_ctx *c)
{
    if (!c) return 0;
    if (c->async_on) (void)halo_async_end(c);                // a failed call left the side-stream scope open: swap back first
    (void)hipStreamSynchronize(c->stream);
    // only the stream this context created is destroyed, whatever c->stream / c->xstream point at
    if (c->own_xstream) { (void)hipStreamSynchronize(c->own_xstream); (void)hipStreamDestroy(c->own_xstream); }
    for (hipEvent_t e : {c->ev_fork, c->ev_join, c->ev_comm, c->ev2d_m, c->ev2d_b}) if (e) (void)hipEventDestroy(e);
    if (c->d_xbuf2) (void)hipFree(c->d_xbuf2);
    for (auto &v : c->tev) for (auto &p : v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &e : c->evpool) (void)hipEventDestroy(e);
    (void)hipFree(c->d_sc); if (c->d_vmixrow) (void)hipFree(c->d_vmixrow); (void)hipFree(c->d_work); if (c->d_xbuf) (void)hipFree(c->d_xbuf); if (c->d_diag) (void)hipFree(c->d_diag);
    comm_release(c);
    delete c;
    return 0;
}
extern "C" int rh_nfast(const rh_ctx *c) { return c->nfast; }
extern "C" int rh_weights(const rh_ctx *c, double *w1, double *w2)
{
    memcpy(w1, c->w1.data(), sizeof(double) * c->w1.size()); memcpy(w2, c->w2.data(), sizeof(double) * c->w2.size());
    return 0;
}
extern "C" int rh_sync(rh_ctx *c) { RH_TRY(halo_async_join(c)); RH_HIP(hipStreamSynchronize(c->stream)); return 0; }

extern "C" void rh_swap_old_new(rh_state *s)
{
    std::swap(s->cons_old, s->cons_new); std::swap(s->xvel_old, s->xvel_new);
    std::swap(s->yvel_old, s->yvel_new); std::swap(s->zvel_old, s->zvel_new);
}
