// How much of the memory system's streaming rate is left when a pointwise kernel reads NR and writes NW separate 1.7-GB arrays at once
// (the stream counts of the path's 3-D kernels: prestep_t3 10 + 6, rhs_t3 10 + 3, rhs_uv_3d 8 + 2, prestep_uv 7 + 4, massflux_omega
// 4 + 3).  Tools only; not part of the library.  Buffers hold random doubles.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { const double *r[12]; double *w[8]; };
template <int NR, int NW> __global__ void rw(Ptrs p, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) v[q] = p.r[q][i];
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NR; q++) s += v[q];
#pragma unroll
    for (int q = 0; q < NW; q++) p.w[q][i] = s + (double)q;
    if (NW == 0 && s == 1.2345e300) p.w[0][i] = s;
}
// the same operand counts walked the way the k-marching kernels walk them: one thread per column (i, j) of a 2054 x 2054 x 50 block,
// 64 x BY threads per block, levels in a loop with the loads of level k + 1 requested before level k is stored
template <int NR, int NW, int BY> __global__ void __launch_bounds__(64 * BY) rwk(Ptrs p, int nx, int ny, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * BY + threadIdx.y;
    if (i >= nx || j >= ny) return;
    const long plane = (long)nx * ny; long o = (long)j * nx + i;
    double v[NR], n[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) v[q] = p.r[q][o];
    for (int k = 0; k < nz; k++) {
        const long on = o + (k + 1 < nz ? plane : 0);
#pragma unroll
        for (int q = 0; q < NR; q++) n[q] = p.r[q][on];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NR; q++) s += v[q];
#pragma unroll
        for (int q = 0; q < NW; q++) p.w[q][o] = s + (double)q;
#pragma unroll
        for (int q = 0; q < NR; q++) v[q] = n[q];
        o = on;
    }
}
// ... and with the blocks of one XCD walking compact PX x PY patches of tiles (rh_common.h: rh_tile_patch), PF levels requested ahead
template <int NR, int NW, int BY, int PX, int PY, int PF> __global__ void __launch_bounds__(64 * BY) rwkp(Ptrs p, int nx, int ny, int nz)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy;
    const unsigned orig = blockIdx.y * gx + blockIdx.x;
    const unsigned qq = n / 8, rr = n % 8, xcd = orig % 8;
    const unsigned t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + orig / 8;
    const unsigned pr = t / (gx * PY), rem = t - pr * gx * PY;
    const unsigned h = min((unsigned)PY, gy - pr * PY);
    const unsigned pc = rem / (PX * h), rem2 = rem - pc * PX * h;
    const unsigned w = min((unsigned)PX, gx - pc * PX);
    const unsigned ly = rem2 / w, lx = rem2 - ly * w;
    const int i = (int)(pc * PX + lx) * 64 + threadIdx.x, j = (int)(pr * PY + ly) * BY + threadIdx.y;
    if (i >= nx || j >= ny) return;
    const long plane = (long)nx * ny; const long o0 = (long)j * nx + i;
    double v[PF][NR];
#pragma unroll
    for (int f = 0; f < PF; f++)
#pragma unroll
        for (int q = 0; q < NR; q++) v[f][q] = p.r[q][o0 + (f < nz ? f : nz - 1) * plane];
    for (int k = 0; k < nz; k += PF) {
#pragma unroll
        for (int f = 0; f < PF; f++) {
            const int kk = k + f; if (kk >= nz) break;
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < NR; q++) s += v[f][q];
            const int kn = kk + PF < nz ? kk + PF : nz - 1;
#pragma unroll
            for (int q = 0; q < NR; q++) v[f][q] = p.r[q][o0 + kn * plane];
#pragma unroll
            for (int q = 0; q < NW; q++) p.w[q][o0 + kk * plane] = s + (double)q;
            if (NW == 0 && s == 1.2345e300) p.w[0][o0] = s;
        }
    }
}
// the same walk along j instead of k: a block owns 64 columns x BY levels and marches over the rows (addresses advance by one row
// pitch per step instead of one 34-MB plane)
template <int NR, int NW, int BY> __global__ void __launch_bounds__(64 * BY) rwj(Ptrs p, int nx, int ny, int nz, int jseg)
{
    const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y * BY + threadIdx.y;
    const int j0 = blockIdx.z * jseg, j1 = min(ny, j0 + jseg);
    if (i >= nx || k >= nz) return;
    long o = (long)k * nx * ny + (long)j0 * nx + i;
    double v[NR], n[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) v[q] = p.r[q][o];
    for (int j = j0; j < j1; j++) {
        const long on = o + (j + 1 < j1 ? nx : 0);
#pragma unroll
        for (int q = 0; q < NR; q++) n[q] = p.r[q][on];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NR; q++) s += v[q];
#pragma unroll
        for (int q = 0; q < NW; q++) p.w[q][o] = s + (double)q;
#pragma unroll
        for (int q = 0; q < NR; q++) v[q] = n[q];
        o = on;
    }
}
__global__ void fill_rand(double *d, long n, unsigned long long seed) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { unsigned long long x = (unsigned long long)i * 6364136223846793005ULL + seed; x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 32; d[i] = 1.0 + (double)(x >> 11) * (1.0 / 9007199254740992.0); } }
int main()
{
    const long n = 2054L * 2054 * 50;
    Ptrs p; double *buf[20];
    for (int q = 0; q < 20; q++) { CK(hipMalloc((void **)&buf[q], n * 8)); fill_rand<<<(n + 255) / 256, 256>>>(buf[q], n, 777ULL + q); }
    CK(hipDeviceSynchronize());
    for (int q = 0; q < 12; q++) p.r[q] = buf[q];
    for (int q = 0; q < 8; q++) p.w[q] = buf[12 + q];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int T = 256; const unsigned G = (unsigned)((n + T - 1) / T);
    auto run = [&](const char *name, int nr, int nw, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        const double nn = (name[8] == 'a' && name[9] == 'l') ? 2048.0 * 2048.0 * 50.0 : (double)n;
        printf("%-28s %2d r + %d w  %8.3f ms  %6.2f TB/s\n", name, nr, nw, best, 8.0 * nn * (nr + nw) / (best * 1e-3) / 1e12); fflush(stdout);
    };
    run("copy", 1, 1, [&] { rw<1, 1><<<G, T>>>(p, n); });
    run("massflux_omega-like", 4, 3, [&] { rw<4, 3><<<G, T>>>(p, n); });
    run("uv3dmix-like", 5, 2, [&] { rw<5, 2><<<G, T>>>(p, n); });
    run("prestep_uv-like", 7, 4, [&] { rw<7, 4><<<G, T>>>(p, n); });
    run("rhs_uv_3d-like", 8, 2, [&] { rw<8, 2><<<G, T>>>(p, n); });
    run("rhs_t3-like", 10, 3, [&] { rw<10, 3><<<G, T>>>(p, n); });
    run("prestep_t3-like", 10, 6, [&] { rw<10, 6><<<G, T>>>(p, n); });
    run("12 r + 8 w", 12, 8, [&] { rw<12, 8><<<G, T>>>(p, n); });
    const dim3 g8((2054 + 63) / 64, (2054 + 7) / 8), b8(64, 8), g16((2054 + 63) / 64, (2054 + 15) / 16), b16(64, 16);
    run("k-march 64x8  rhs_uv-like", 8, 2, [&] { rwk<8, 2, 8><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8  rhs_t3-like", 10, 3, [&] { rwk<10, 3, 8><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8  prestep_t3-like", 10, 6, [&] { rwk<10, 6, 8><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x16 prestep_t3-like", 10, 6, [&] { rwk<10, 6, 16><<<g16, b16>>>(p, 2054, 2054, 50); });
    run("k-march 64x8  uv3dmix-like", 5, 2, [&] { rwk<5, 2, 8><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8 patch 8x4 pf1 pt3", 10, 6, [&] { rwkp<10, 6, 8, 8, 4, 1><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8 patch 8x4 pf2 pt3", 10, 6, [&] { rwkp<10, 6, 8, 8, 4, 2><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8 patch 16x2 pf1 pt3", 10, 6, [&] { rwkp<10, 6, 8, 16, 2, 1><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8 patch 33x1 pf1 pt3", 10, 6, [&] { rwkp<10, 6, 8, 33, 1, 1><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8 patch 8x4 pf1 rhsuv", 8, 2, [&] { rwkp<8, 2, 8, 8, 4, 1><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8 patch 8x4 pf2 rhsuv", 8, 2, [&] { rwkp<8, 2, 8, 8, 4, 2><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march 64x8 patch 8x4 pf4 rhsuv", 8, 2, [&] { rwkp<8, 2, 8, 8, 4, 4><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march patch 8x4 read only", 10, 0, [&] { rwkp<10, 0, 8, 8, 4, 1><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march patch 8x4 1 r + 6 w", 1, 6, [&] { rwkp<1, 6, 8, 8, 4, 1><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("k-march patch 8x4 4 r + 1 w", 4, 1, [&] { rwkp<4, 1, 8, 8, 4, 1><<<g8, b8>>>(p, 2054, 2054, 50); });
    run("flat 10 r + 0 w", 10, 0, [&] { rw<10, 0><<<G, T>>>(p, n); });
    run("flat 1 r + 6 w", 1, 6, [&] { rw<1, 6><<<G, T>>>(p, n); });
    { const dim3 ga(2048 / 64, 2048 / 8);       // rows of 2048 doubles: every wave's 512-B piece starts on a 512-B boundary
      run("k-march aligned 1 r + 6 w", 1, 6, [&] { rwkp<1, 6, 8, 8, 4, 1><<<ga, b8>>>(p, 2048, 2048, 50); });
      run("k-march aligned 10 r + 6 w", 10, 6, [&] { rwkp<10, 6, 8, 8, 4, 1><<<ga, b8>>>(p, 2048, 2048, 50); });
      run("k-march aligned 8 r + 2 w", 8, 2, [&] { rwkp<8, 2, 8, 8, 4, 1><<<ga, b8>>>(p, 2048, 2048, 50); }); }
    { const dim3 ga(2048 / 64, 2048 / 8);       // aligned pitch, but every piece starts 24 B into its line (the ghost offset of a dense array)
      Ptrs q = p; for (int z = 0; z < 12; z++) q.r[z] = p.r[z] + 3; for (int z = 0; z < 8; z++) q.w[z] = p.w[z] + 3;
      run("k-march al.pitch+24B 1 r + 6 w", 1, 6, [&] { rwkp<1, 6, 8, 8, 4, 1><<<ga, b8>>>(q, 2048, 2048, 50); });
      run("k-march al.pitch+24B 10 r + 6 w", 10, 6, [&] { rwkp<10, 6, 8, 8, 4, 1><<<ga, b8>>>(q, 2048, 2048, 50); });
      for (int z = 0; z < 8; z++) q.w[z] = p.w[z] + 4;       // 32-B aligned stores
      run("k-march al.pitch+32B 1 r + 6 w", 1, 6, [&] { rwkp<1, 6, 8, 8, 4, 1><<<ga, b8>>>(q, 2048, 2048, 50); });
      for (int z = 0; z < 8; z++) q.w[z] = p.w[z] + 16;      // 128-B aligned stores
      run("k-march al.pitch+128B 1 r + 6 w", 1, 6, [&] { rwkp<1, 6, 8, 8, 4, 1><<<ga, b8>>>(q, 2048, 2048, 50); }); }
    { const int jseg = 64; const dim3 gj((2054 + 63) / 64, (50 + 7) / 8, (2054 + jseg - 1) / jseg);
      run("j-march 64x8lev 64 rows pt3", 10, 6, [&] { rwj<10, 6, 8><<<gj, b8>>>(p, 2054, 2054, 50, jseg); });
      run("j-march 64x8lev 64 rows rhsuv", 8, 2, [&] { rwj<8, 2, 8><<<gj, b8>>>(p, 2054, 2054, 50, jseg); }); }
    { const int jseg = 256; const dim3 gj((2054 + 63) / 64, (50 + 7) / 8, (2054 + jseg - 1) / jseg);
      run("j-march 64x8lev 256 rows pt3", 10, 6, [&] { rwj<10, 6, 8><<<gj, b8>>>(p, 2054, 2054, 50, jseg); }); }
    return 0;
}
