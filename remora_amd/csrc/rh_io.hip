// rh_io.hip -- AMReX-native plotfile output of the state, so that results can be read by the reference's own tooling
// (amrex_fcompare, yt, VisIt): REMORA::WritePlotFile / WriteMultiLevelPlotfileWithBathymetry / WriteGenericPlotfileHeaderWithBathymetry,
// Source/IO/REMORA_Plotfile.cpp:141-605, for one level.  Layout:
//   <dir>/Header                       text: variables, geometry, one RealBox per box, the nodal "Nu_nd" block
//   <dir>/Level_0/Cell_H, Nu_nd_H      VisMF headers: box list, "FabOnDisk: <file> <offset>", per-box minima and maxima
//   <dir>/Level_0/Cell_D_nnnnn         "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))<box> <ncomp>\n" + native doubles
//   <dir>/Level_0/Nu_nd_D_nnnnn        the same for the 3-component nodal mesh displacement (0, 0, z_phys_nd - k dz)
// The variables are gathered on the device (two kernels), copied to the host once and written with stdio; only this file and the
// diagnostics synchronise.  Every rank writes its own data file; the headers take the list of all boxes (rank 0, after a gather).
#include "rh_common.h"
#include <sstream>
#include <algorithm>
#include <sys/stat.h>
#include <cerrno>

// nodal mesh height: z_phys_nd of Utils/REMORA_DepthStretchTransform.H:204-236 (average of the four w-columns around the node; the
// z_w array carries three ghost columns, so the clamped branch of :233-236 is never taken on the nodes of a valid box) minus the
// uniform-grid height k dz (REMORA_Plotfile.cpp:320-329)
__global__ void k_plot_nd(B3 nb, A4 z_w, double *__restrict__ out, double dz)
{
    RH_IJ(nb); if (!inb2(nb, i, j)) return;
    const int k = nb.lo2 + RH_KZ;
    const long nx = nb.hi0 - nb.lo0 + 1, ny = nb.hi1 - nb.lo1 + 1, nz = nb.hi2 - nb.lo2 + 1, n3 = nx * ny * nz;
    const long q = (long)(i - nb.lo0) + (long)(j - nb.lo1) * nx + (long)(k - nb.lo2) * nx * ny;
    double z = 0.25 * (z_w(i, j, k) + z_w(i + 1, j, k) + z_w(i, j + 1, k) + z_w(i + 1, j + 1, k));
    z -= k * dz;
    out[q] = 0.0; out[n3 + q] = 0.0; out[2 * n3 + q] = z;
}

static int make_dir(const std::string &d)
{
    if (mkdir(d.c_str(), 0777) != 0 && errno != EEXIST) return rh_fail("plotfile: cannot create directory %s: %s", d.c_str(), strerror(errno));
    return 0;
}
static std::string box_str(const rh_box &b, int nodal)
{
    char s[160];
    snprintf(s, sizeof s, "((%d,%d,%d) (%d,%d,%d) (%d,%d,%d))", b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2], nodal, nodal, nodal);
    return s;
}
// one FAB record written to `path` (truncated first: every rank owns its data file and writes exactly one record per plotfile, so
// a repeated write into an existing directory must not grow the file); returns the offset the record starts at
static int write_fab(const std::string &path, const rh_box &b, int nodal, int ncomp, const double *data, long *offset)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return rh_fail("plotfile: cannot open %s: %s", path.c_str(), strerror(errno));
    fseek(f, 0, SEEK_END);
    *offset = ftell(f);
    fprintf(f, "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))%s %d\n", box_str(b, nodal).c_str(), ncomp);
    const size_t n = (size_t)bx_npts(b) * ncomp;
    const size_t w = fwrite(data, sizeof(double), n, f);
    if (fclose(f) != 0 || w != n) return rh_fail("plotfile: short write to %s", path.c_str());
    return 0;
}
static void min_max(const double *d, long n3, int ncomp, double *mn, double *mx)
{
    for (int c = 0; c < ncomp; c++) {
        double a = d[c * n3], b = a;
        for (long q = 1; q < n3; q++) { const double v = d[c * n3 + q]; a = v < a ? v : a; b = v > b ? v : b; }
        mn[c] = a; mx[c] = b;
    }
}

extern "C" int rh_plotfile_write_data(rh_ctx *c, rh_state *s, const char *dir, int with_scalar, int file_index, rh_plot_box *out)
{
    if (!dir || !out) return rh_fail("rh_plotfile_write_data: null argument");
    if (file_index < 0 || file_index > 99999) return rh_fail("rh_plotfile_write_data: file_index out of range");
    const rh_box bx = c->bx;
    rh_box nb = bx; for (int d = 0; d < 3; d++) nb.hi[d] += 1;                    // surroundingNodes in all directions
    RH_NEED(s->z_w, bx_grow(bx_nodal(bx, 2), 1, 1, 0));
    const int nc = with_scalar ? 6 : 5;
    const long n3 = bx_npts(bx), m3 = bx_npts(nb);
    double *dbuf = nullptr;
    RH_HIP(hipMalloc((void **)&dbuf, sizeof(double) * (size_t)(nc * n3 + 3 * m3)));
    int rc = rh_plot_vars(c, s, with_scalar, dbuf);
    if (!rc) {
        const double dz = (c->prm.prob_hi[2] - c->prm.prob_lo[2]) / (double)c->nz;
        k_plot_nd<<<grid3(nb), rh_block(), 0, c->stream>>>(mkB(nb), mkA(s->z_w), dbuf + nc * n3, dz);
        if (hipGetLastError() != hipSuccess) rc = rh_fail("rh_plotfile_write_data: launch failed");
    }
    std::vector<double> h((size_t)(nc * n3 + 3 * m3));
    if (!rc && hipMemcpyAsync(h.data(), dbuf, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = rh_fail("rh_plotfile_write_data: copy failed");
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = rh_fail("rh_plotfile_write_data: synchronise failed");
    (void)hipFree(dbuf);
    if (rc) return rc;
    const std::string d0 = dir, d1 = d0 + "/Level_0";
    RH_TRY(make_dir(d0)); RH_TRY(make_dir(d1));
    char name[64];
    memset(out, 0, sizeof *out);
    out->box = bx; out->file_index = file_index;
    snprintf(name, sizeof name, "/Cell_D_%05d", file_index);
    RH_TRY(write_fab(d1 + name, bx, 0, nc, h.data(), &out->cell_offset));
    snprintf(name, sizeof name, "/Nu_nd_D_%05d", file_index);
    RH_TRY(write_fab(d1 + name, nb, 1, 3, h.data() + nc * n3, &out->nd_offset));
    min_max(h.data(), n3, nc, out->cell_min, out->cell_max);
    min_max(h.data() + nc * n3, m3, 3, out->nd_min, out->nd_max);
    return 0;
}

// VisMF header (AMReX VisMF::Header, version 1, one FAB per box): what Cell_H / Nu_nd_H hold in the reference's gold files
static int write_vismf_header(const std::string &path, const char *prefix, int ncomp, int nodal, int nboxes, const rh_plot_box *B)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return rh_fail("plotfile: cannot open %s: %s", path.c_str(), strerror(errno));
    fprintf(f, "1\n1\n%d\n0\n(%d 0\n", ncomp, nboxes);
    for (int b = 0; b < nboxes; b++) {
        rh_box x = B[b].box; if (nodal) for (int d = 0; d < 3; d++) x.hi[d] += 1;
        fprintf(f, "%s\n", box_str(x, nodal).c_str());
    }
    fprintf(f, ")\n%d\n", nboxes);
    for (int b = 0; b < nboxes; b++) fprintf(f, "FabOnDisk: %s_D_%05d %ld\n", prefix, B[b].file_index, nodal ? B[b].nd_offset : B[b].cell_offset);
    for (int pass = 0; pass < 2; pass++) {
        fprintf(f, "\n%d,%d\n", nboxes, ncomp);
        for (int b = 0; b < nboxes; b++) {
            const double *v = nodal ? (pass ? B[b].nd_max : B[b].nd_min) : (pass ? B[b].cell_max : B[b].cell_min);
            for (int n = 0; n < ncomp; n++) fprintf(f, "%.16e,", v[n]);
            fprintf(f, "\n");
        }
    }
    fprintf(f, "\n");
    return fclose(f) == 0 ? 0 : rh_fail("plotfile: write to %s failed", path.c_str());
}

extern "C" int rh_plotfile_write_headers(const rh_params *P, const char *dir, int with_scalar, int step, double time, int nboxes, const rh_plot_box *B)
{
    if (!P || !dir || !B || nboxes < 1) return rh_fail("rh_plotfile_write_headers: bad argument");
    const std::string d0 = dir, d1 = d0 + "/Level_0";
    RH_TRY(make_dir(d0)); RH_TRY(make_dir(d1));
    const int nc = with_scalar ? 6 : 5;
    double dx[3];
    for (int d = 0; d < 3; d++) dx[d] = (P->prob_hi[d] - P->prob_lo[d]) / (double)(P->domain_hi[d] - P->domain_lo[d] + 1);
    std::ostringstream H; H.precision(17);                                        // REMORA_Plotfile.cpp:520-605
    H << "HyperCLaw-V1.1" << '\n' << nc << '\n';
    H << "temp\nsalt\n"; if (with_scalar) H << "scalar\n";
    H << "x_velocity\ny_velocity\nz_velocity\n";
    H << 3 << '\n' << time << '\n' << 0 << '\n';
    for (int d = 0; d < 3; d++) H << P->prob_lo[d] << ' ';
    H << '\n';
    for (int d = 0; d < 3; d++) H << P->prob_hi[d] << ' ';
    H << '\n' << '\n';                                                            // (no refinement ratios on one level)
    rh_box dom; for (int d = 0; d < 3; d++) { dom.lo[d] = P->domain_lo[d]; dom.hi[d] = P->domain_hi[d]; }
    H << box_str(dom, 0) << ' ' << '\n' << step << ' ' << '\n';
    for (int d = 0; d < 3; d++) H << dx[d] << ' ';
    H << '\n' << 0 << '\n' << "0\n";
    H << 0 << ' ' << nboxes << ' ' << time << '\n' << step << '\n';
    for (int b = 0; b < nboxes; b++)
        for (int d = 0; d < 3; d++) {                                             // amrex::RealBox(b, dx, prob_lo), box shifted to the domain's origin
            const double lo = P->prob_lo[d] + dx[d] * (double)(B[b].box.lo[d] - P->domain_lo[d]);
            const double hi = P->prob_lo[d] + dx[d] * (double)(B[b].box.hi[d] - P->domain_lo[d] + 1);
            H << lo << ' ' << hi << '\n';
        }
    H << "Level_0/Cell" << '\n' << "1\n3\namrexvec_nu_x\namrexvec_nu_y\namrexvec_nu_z\n" << "Level_0/Nu_nd" << '\n';
    FILE *f = fopen((d0 + "/Header").c_str(), "w");
    if (!f) return rh_fail("plotfile: cannot open %s/Header: %s", dir, strerror(errno));
    const std::string hs = H.str();
    const size_t w = fwrite(hs.data(), 1, hs.size(), f);
    if (fclose(f) != 0 || w != hs.size()) return rh_fail("plotfile: write to %s/Header failed", dir);
    RH_TRY(write_vismf_header(d1 + "/Cell_H", "Cell", nc, 0, nboxes, B));
    return write_vismf_header(d1 + "/Nu_nd_H", "Nu_nd", 3, 1, nboxes, B);
}

extern "C" int rh_write_plotfile(rh_ctx *c, rh_state *s, const char *dir, int with_scalar, int step, double time)
{
    if (c->prm.nranks[0] * c->prm.nranks[1] > 1)
        return rh_fail("rh_write_plotfile: decomposed runs gather their rh_plot_box records and call rh_plotfile_write_headers on one rank");
    rh_plot_box b;
    RH_TRY(rh_plotfile_write_data(c, s, dir, with_scalar, 0, &b));
    return rh_plotfile_write_headers(&c->prm, dir, with_scalar, step, time, 1, &b);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Checkpoint / restart: REMORA::WriteCheckpointFile / ReadCheckpointFile, Source/IO/REMORA_Checkpoint.cpp:15-245,248-470.
// The reference writes every MultiFab WITH its ghost cells (the copies keep nGrowVect, :115-202), one VisMF per array under
// <dir>/Level_0/<Name>_{H,D_00000}, and a text Header (:57-106).  A native rh_array4 is already one FAB in that memory order, so a
// record is the array as it stands.  Single box per call; a decomposed run passes its rank as file_index and a directory per rank
// or writes the headers itself (the per-array headers list one box).
struct ChkField { const char *name; rh_array4 rh_state::*m; int ix0, ix1, ix2; int gls_only; };
static const ChkField g_chk[] = {
    {"Cell", &rh_state::cons_new, 0, 0, 0, 0}, {"Cell_old", &rh_state::cons_old, 0, 0, 0, 0},
    {"XFace", &rh_state::xvel_new, 1, 0, 0, 0}, {"XFace_old", &rh_state::xvel_old, 1, 0, 0, 0},
    {"YFace", &rh_state::yvel_new, 0, 1, 0, 0}, {"YFace_old", &rh_state::yvel_old, 0, 1, 0, 0},
    {"ZFace", &rh_state::zvel_new, 0, 0, 1, 0}, {"ZFace_old", &rh_state::zvel_old, 0, 0, 1, 0},
    {"XRHS", &rh_state::ru, 1, 0, 0, 0}, {"YRHS", &rh_state::rv, 0, 1, 0, 0},
    {"XBar", &rh_state::ubar, 1, 0, 0, 0}, {"YBar", &rh_state::vbar, 0, 1, 0, 0},
    {"XRHS2d", &rh_state::ru2d, 1, 0, 0, 0}, {"YRHS2d", &rh_state::rv2d, 0, 1, 0, 0},
    {"Mskr", &rh_state::mskr, 0, 0, 0, 0}, {"Msku", &rh_state::msku, 1, 0, 0, 0}, {"Mskv", &rh_state::mskv, 0, 1, 0, 0},
    {"rdrag", &rh_state::rdrag, 0, 0, 0, 0},
    {"DU_avg1", &rh_state::DU_avg1, 1, 0, 0, 0}, {"DU_avg2", &rh_state::DU_avg2, 1, 0, 0, 0},
    {"DV_avg1", &rh_state::DV_avg1, 0, 1, 0, 0}, {"DV_avg2", &rh_state::DV_avg2, 0, 1, 0, 0},
    {"zeta", &rh_state::zeta, 0, 0, 0, 0}, {"Zt_avg1", &rh_state::Zt_avg1, 0, 0, 0, 0}, {"h", &rh_state::h, 0, 0, 0, 0},
    {"tke", &rh_state::tke, 0, 0, 1, 1}, {"gls", &rh_state::gls, 0, 0, 1, 1}, {"Lscale", &rh_state::Lscale, 0, 0, 1, 1},
    {"Akk", &rh_state::Akk, 0, 0, 1, 1}, {"Akp", &rh_state::Akp, 0, 0, 1, 1},
    {"Akv", &rh_state::Akv, 0, 0, 1, 0}, {"Akt", &rh_state::Akt, 0, 0, 1, 0},
};
// shape of the reference's GLS arrays on this rank's box: w-points, ghost widths (NGROW, NGROW, 0) (make_new_level.cpp:499-503)
static rh_array4 gls_shape(const rh_ctx *c, int ncomp)
{
    rh_array4 a; a.p = nullptr; a.ncomp = ncomp;
    const rh_box b = bx_grow(bx_nodal(c->bx, 2), RH_NGROW, RH_NGROW, 0);
    for (int d = 0; d < 3; d++) { a.lo[d] = b.lo[d]; a.hi[d] = b.hi[d]; }
    return a;
}
static std::string box_str_ix(const rh_box &b, int ix0, int ix1, int ix2)
{
    char s[160];
    snprintf(s, sizeof s, "((%d,%d,%d) (%d,%d,%d) (%d,%d,%d))", b.lo[0], b.lo[1], b.lo[2], b.hi[0], b.hi[1], b.hi[2], ix0, ix1, ix2);
    return s;
}

extern "C" int rh_write_checkpoint(rh_ctx *c, rh_state *s, const char *dir, int step, double time)
{
    if (!dir) return rh_fail("rh_write_checkpoint: null directory");
    const std::string d0 = dir, d1 = d0 + "/Level_0";
    RH_TRY(make_dir(d0)); RH_TRY(make_dir(d1));
    RH_HIP(hipStreamSynchronize(c->stream));
    std::vector<double> h;
    const bool have_gls = c->prm.vmix_mode == RH_VMIX_GLS;
    for (const ChkField &f : g_chk) {
        rh_array4 a = s->*(f.m);
        // The reference writes tke / gls / Lscale / Akk / Akp whatever the mixing scheme (:196-200) and reads them back
        // unconditionally (:448-452).  A state without the closure has no such arrays: they are written in the reference's shape
        // (w-points, ghost widths (2,2,0), 3/3/1/1/1 components, make_new_level.cpp:499-503) holding what init leaves there
        // (gls_Kmin, gls_Pmin, 0, Akk_bak, Akp_bak: make_new_level.cpp:517-521; defaults of REMORA_DataStruct.H:430-443).
        const bool synth = f.gls_only && !have_gls;
        double fill = 0.0;
        if (synth) {
            const rh_gls_params &G = c->prm.gls;
            const bool tk = !strcmp(f.name, "tke"), gl = !strcmp(f.name, "gls");
            a = gls_shape(c, (tk || gl) ? 3 : 1);
            if (tk) fill = G.gls_Kmin > 0.0 ? G.gls_Kmin : 7.6e-6;
            else if (gl) fill = G.gls_Pmin > 0.0 ? G.gls_Pmin : 1.0e-12;
            else if (!strcmp(f.name, "Akk")) fill = G.Akk_bak > 0.0 ? G.Akk_bak : 5.0e-6;
            else if (!strcmp(f.name, "Akp")) fill = G.Akp_bak > 0.0 ? G.Akp_bak : 5.0e-6;
        } else if (!a.p) return rh_fail("rh_write_checkpoint: state array %s is not bound", f.name);
        const MFV m = mfv(c, a, f.ix0, f.ix1, f.ix2);
        const long n = a4_size(a), n3 = n / a.ncomp;
        h.resize((size_t)n);
        if (synth) std::fill(h.begin(), h.end(), fill);
        else RH_HIP(hipMemcpy(h.data(), a.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
        const std::string base = d1 + "/" + f.name;
        FILE *fd = fopen((base + "_D_00000").c_str(), "wb");
        if (!fd) return rh_fail("rh_write_checkpoint: cannot open %s_D_00000: %s", base.c_str(), strerror(errno));
        fprintf(fd, "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))%s %d\n", box_str_ix(bx_of(a), f.ix0, f.ix1, f.ix2).c_str(), a.ncomp);
        const size_t w = fwrite(h.data(), sizeof(double), (size_t)n, fd);
        if (fclose(fd) != 0 || w != (size_t)n) return rh_fail("rh_write_checkpoint: short write to %s_D_00000", base.c_str());
        // VisMF header: version, how, ncomp, ghost widths, box list (valid boxes), FabOnDisk, minima, maxima (over the whole FAB)
        FILE *fh = fopen((base + "_H").c_str(), "w");
        if (!fh) return rh_fail("rh_write_checkpoint: cannot open %s_H", base.c_str());
        fprintf(fh, "1\n1\n%d\n", a.ncomp);
        if (m.ng[0] == m.ng[1] && m.ng[1] == m.ng[2]) fprintf(fh, "%d\n", m.ng[0]); else fprintf(fh, "(%d,%d,%d)\n", m.ng[0], m.ng[1], m.ng[2]);
        fprintf(fh, "(1 0\n%s\n)\n1\nFabOnDisk: %s_D_00000 0\n", box_str_ix(m.valid, f.ix0, f.ix1, f.ix2).c_str(), f.name);
        std::vector<double> mn(a.ncomp), mx(a.ncomp);
        min_max(h.data(), n3, a.ncomp, mn.data(), mx.data());
        for (int pass = 0; pass < 2; pass++) {
            fprintf(fh, "\n1,%d\n", a.ncomp);
            for (int q = 0; q < a.ncomp; q++) fprintf(fh, "%.16e,", pass ? mx[q] : mn[q]);
            fprintf(fh, "\n");
        }
        fprintf(fh, "\n");
        if (fclose(fh) != 0) return rh_fail("rh_write_checkpoint: write to %s_H failed", base.c_str());
    }
    std::ostringstream H; H.precision(17);                                        // REMORA_Checkpoint.cpp:57-106
    H << "Checkpoint file for REMORA\n" << 0 << "\n" << RH_NCONS << "\n" << 1 << "\n" << 1 << "\n" << 1 << "\n" << 2 << "\n" << 2 << "\n" << 3 << "\n" << 3 << "\n";
    H << step << " \n" << c->prm.dt << " \n" << time << " \n";
    H << "(1 0\n" << box_str_ix(c->bx, 0, 0, 0) << "\n)\n" << '\n';
    FILE *f = fopen((d0 + "/Header").c_str(), "w");
    if (!f) return rh_fail("rh_write_checkpoint: cannot open %s/Header", dir);
    const std::string hs = H.str();
    const size_t w = fwrite(hs.data(), 1, hs.size(), f);
    if (fclose(f) != 0 || w != hs.size()) return rh_fail("rh_write_checkpoint: write to %s/Header failed", dir);
    return 0;
}

// Restart: the state must already hold what InitData builds without a checkpoint (metrics, Coriolis, mixing coefficients, masks:
// rh_init_state or the host's own init); this call overwrites the checkpointed arrays, ghost cells included, and rebuilds the
// vertical grid from the restored h and Zt_avg1 (REMORA_Checkpoint.cpp:248-470, then stretch_transform as in REMORA::restart).
extern "C" int rh_read_checkpoint(rh_ctx *c, rh_state *s, const char *dir, int *step, double *time)
{
    if (!dir) return rh_fail("rh_read_checkpoint: null directory");
    c->vmix_akv_done = c->vmix_akt_done = nullptr;          // Akv / Akt come from the file: the next step forms the constants again
    const std::string d0 = dir, d1 = d0 + "/Level_0";
    {
        FILE *f = fopen((d0 + "/Header").c_str(), "r");
        if (!f) return rh_fail("rh_read_checkpoint: cannot open %s/Header: %s", dir, strerror(errno));
        char line[256]; int finest = -1, nc[8], st = 0; double dt = 0, t = 0;
        bool ok = fgets(line, sizeof line, f) && strncmp(line, "Checkpoint file for REMORA", 26) == 0 && fscanf(f, "%d", &finest) == 1;
        for (int q = 0; ok && q < 8; q++) ok = fscanf(f, "%d", &nc[q]) == 1;
        ok = ok && fscanf(f, "%d %lf %lf", &st, &dt, &t) == 3;
        fclose(f);
        if (!ok) return rh_fail("rh_read_checkpoint: %s/Header is not a REMORA checkpoint header", dir);
        if (finest != 0) return rh_fail("rh_read_checkpoint: %d levels in the checkpoint; one level is supported", finest + 1);
        if (nc[0] != RH_NCONS) return rh_fail("rh_read_checkpoint: checkpoint holds %d tracers, this build %d", nc[0], RH_NCONS);
        if (step) *step = st;
        if (time) *time = t;
    }
    std::vector<double> h;
    for (const ChkField &f : g_chk) {
        const rh_array4 &a = s->*(f.m);
        // a state without the closure has nowhere to put tke / gls / Lscale / Akk / Akp (nothing on its path reads them)
        if (f.gls_only && c->prm.vmix_mode != RH_VMIX_GLS) continue;
        if (!a.p) return rh_fail("rh_read_checkpoint: state array %s is not bound", f.name);
        const std::string base = d1 + "/" + f.name;
        long off = -1; char fname[128] = "";
        {
            FILE *fh = fopen((base + "_H").c_str(), "r");
            if (!fh) return rh_fail("rh_read_checkpoint: cannot open %s_H: %s", base.c_str(), strerror(errno));
            char line[512];
            while (fgets(line, sizeof line, fh)) if (sscanf(line, "FabOnDisk: %127s %ld", fname, &off) == 2) break;
            fclose(fh);
            if (off < 0) return rh_fail("rh_read_checkpoint: no FabOnDisk record in %s_H", base.c_str());
        }
        FILE *fd = fopen((d1 + "/" + fname).c_str(), "rb");
        if (!fd) return rh_fail("rh_read_checkpoint: cannot open %s/%s", d1.c_str(), fname);
        fseek(fd, off, SEEK_SET);
        char hdr[512];
        if (!fgets(hdr, sizeof hdr, fd)) { fclose(fd); return rh_fail("rh_read_checkpoint: %s: no FAB header", fname); }
        const char *p = strstr(hdr, ")))");                                       // end of the real descriptor
        int lo[3], hi[3], ix[3], ncomp = 0;
        if (!p || sscanf(p + 3, "((%d,%d,%d) (%d,%d,%d) (%d,%d,%d)) %d", &lo[0], &lo[1], &lo[2], &hi[0], &hi[1], &hi[2], &ix[0], &ix[1], &ix[2], &ncomp) != 10 ||
            strncmp(hdr, "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))", 58) != 0) {
            fclose(fd); return rh_fail("rh_read_checkpoint: %s: not a native little-endian double FAB", fname);
        }
        bool same = ncomp == a.ncomp;
        for (int d = 0; d < 3; d++) same = same && lo[d] == a.lo[d] && hi[d] == a.hi[d];
        if (!same) { fclose(fd); return rh_fail("rh_read_checkpoint: %s holds box (%d,%d,%d)-(%d,%d,%d) x %d, the state array (%d,%d,%d)-(%d,%d,%d) x %d (same decomposition and ghost widths are required)",
                                                fname, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], ncomp, a.lo[0], a.lo[1], a.lo[2], a.hi[0], a.hi[1], a.hi[2], a.ncomp); }
        const long n = a4_size(a);
        h.resize((size_t)n);
        const size_t r = fread(h.data(), sizeof(double), (size_t)n, fd);
        fclose(fd);
        if (r != (size_t)n) return rh_fail("rh_read_checkpoint: %s is truncated", fname);
        RH_HIP(hipMemcpy(a.p, h.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    }
    return rh_stretch_transform(c, s);
}
