// Two questions about the memory path of the k-marching tiled kernels (tools only; not part of the library):
//  (1) what a pointwise kernel with the path's stream counts reaches with 16 bytes per lane (two i-points per thread) instead of 8,
//      on rows that start 24 B into a 128-B line like the ghost offset of a dense array;
//  (2) whether filling the LDS tiles of a k-marching block (64 x 8 points + halo 2, NR arrays, double-buffered over k) by LDS-DMA
//      (global_load_lds_dwordx4: no staging registers, no ds_write pass) moves more bytes per second than staging through VGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Ptrs { const double *r[12]; double *w[8]; };
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int NR, int NW> __global__ void rw1(Ptrs p, long n)
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
// two consecutive doubles per thread, one 16-byte access each way (the pointers are only 8-byte aligned: packed struct)
struct __attribute__((packed, aligned(8))) D2 { double a, b; };
template <int NR, int NW> __global__ void rw2(Ptrs p, long n)
{
    const long i = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n) return;
    D2 v[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) v[q] = *(const D2 *)(p.r[q] + i);
    double s = 0.0, t = 0.0;
#pragma unroll
    for (int q = 0; q < NR; q++) { s += v[q].a; t += v[q].b; }
#pragma unroll
    for (int q = 0; q < NW; q++) { D2 o; o.a = s + (double)q; o.b = t + (double)q; *(D2 *)(p.w[q] + i) = o; }
    if (NW == 0 && s + t == 1.2345e300) p.w[0][i] = s;
}

// ---- k-marching tile probe.  Block = 64 x 8 threads, tile = 68 x 12 doubles per array and level, two LDS buffers per array.
#define TW 68
#define TH 12
#define TN (TW * TH)
__device__ __forceinline__ void xcd_tile(int &tx, int &ty)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy, orig = blockIdx.y * gx + blockIdx.x;
    const unsigned qq = n / 8, rr = n % 8, xcd = orig % 8;
    const unsigned t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + orig / 8;
    const unsigned PX = 8, PY = 4;
    const unsigned pr = t / (gx * PY), rem = t - pr * gx * PY;
    const unsigned h = min(PY, gy - pr * PY);
    const unsigned pc = rem / (PX * h), rem2 = rem - pc * PX * h;
    const unsigned w = min(PX, gx - pc * PX);
    const unsigned ly = rem2 / w, lx = rem2 - ly * w;
    tx = (int)(pc * PX + lx); ty = (int)(pr * PY + ly);
}
// MODE 0: tile elements staged through registers (two per thread and array, linear tile order), written to LDS after the compute
// MODE 1: LDS-DMA, 16 bytes per lane: wave w fills tile doubles [128 w, 128 w + 127]
template <int NR, int NW, int MODE> __global__ void __launch_bounds__(512) tile_march(Ptrs p, int nx, int ny, int nz, int pitch)
{
    extern __shared__ double lds[];                                   // [2][NR][TN]
    int tx, ty; xcd_tile(tx, ty);
    const int i0 = tx * 64, j0 = ty * 8;                              // tile origin (the arrays carry a 3-cell frame: element (0,0) is interior)
    const int tid = threadIdx.y * 64 + threadIdx.x, lane = threadIdx.x, wv = threadIdx.y;
    const long plane = (long)pitch * (ny + 6);
    // tile element e -> global offset of level 0 (clamped to the array)
    auto goff = [&](int e) -> long {
        const int r = e / TW, c = e - r * TW;
        int gi = i0 - 2 + c, gj = j0 - 2 + r;
        gi = min(max(gi, -3), nx + 2); gj = min(max(gj, -3), ny + 2);
        return (long)(gj + 3) * pitch + (gi + 3);
    };
    const int e0 = tid, e1 = tid + 512;
    const long o0 = goff(e0), o1 = goff(e1 < TN ? e1 : TN - 1);
    const int pe = 2 * (wv * 64 + lane);                              // DMA: this lane's pair
    const long od = goff(pe < TN ? pe : TN - 2);
    const bool dma_on = pe < TN;
    const int i = i0 + threadIdx.x, j = j0 + threadIdx.y;
    const bool in = i < nx && j < ny;
    const long oc = (long)(j + 3) * pitch + (i + 3);
    const int lc = (threadIdx.y + 2) * TW + threadIdx.x + 2;
    auto fill_dma = [&](int buf, int k) {
        if (dma_on) {
#pragma unroll
            for (int a = 0; a < NR; a++)
                __builtin_amdgcn_global_load_lds((gptr_t *)(p.r[a] + od + k * plane), (lptr_t *)(lds + (buf * NR + a) * TN + wv * 128), 16, 0, 0);
        }
    };
    double v0[NR], v1[NR];
    if (MODE == 0) {
#pragma unroll
        for (int a = 0; a < NR; a++) { lds[a * TN + e0] = p.r[a][o0]; if (e1 < TN) lds[a * TN + e1] = p.r[a][o1]; }
    } else fill_dma(0, 0);
    __syncthreads();
    for (int k = 0; k < nz; k++) {
        const int cur = k & 1, nxt = cur ^ 1, kn = k + 1 < nz ? k + 1 : k;
        if (MODE == 0) {
#pragma unroll
            for (int a = 0; a < NR; a++) { v0[a] = p.r[a][o0 + kn * plane]; v1[a] = p.r[a][o1 + kn * plane]; }
        } else fill_dma(nxt, kn);
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < NR; a++) {
            const double *t = lds + (cur * NR + a) * TN + lc;
            s += t[0] + t[-2] + t[2] + t[-2 * TW] + t[2 * TW];
        }
        if (in) {
#pragma unroll
            for (int q = 0; q < NW; q++) p.w[q][oc + k * plane] = s + (double)q;
        }
        if (MODE == 0) {
#pragma unroll
            for (int a = 0; a < NR; a++) { lds[(nxt * NR + a) * TN + e0] = v0[a]; if (e1 < TN) lds[(nxt * NR + a) * TN + e1] = v1[a]; }
        }
        __syncthreads();
    }
}
__global__ void fill_rand(double *d, long n, unsigned long long seed) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { unsigned long long x = (unsigned long long)i * 6364136223846793005ULL + seed; x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 32; d[i] = 1.0 + (double)(x >> 11) * (1.0 / 9007199254740992.0); } }
__global__ void diff(const double *a, const double *b, long n, unsigned long long *cnt) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n && a[i] != b[i]) atomicAdd(cnt, 1ULL); }
int main()
{
    const int nx = 2048, ny = 2048, nz = 50, pitch = 2054;
    const long n = (long)pitch * (ny + 6) * nz;
    Ptrs p; double *buf[20];
    for (int q = 0; q < 20; q++) { CK(hipMalloc((void **)&buf[q], n * 8 + 4096)); fill_rand<<<(n + 255) / 256, 256>>>(buf[q], n, 777ULL + q); }
    CK(hipDeviceSynchronize());
    for (int q = 0; q < 12; q++) p.r[q] = buf[q];
    for (int q = 0; q < 8; q++) p.w[q] = buf[12 + q];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int nr, int nw, double nn, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-40s %2d r + %d w  %8.3f ms  %6.2f TB/s\n", name, nr, nw, best, 8.0 * nn * (nr + nw) / (best * 1e-3) / 1e12); fflush(stdout);
    };
    const int T = 256; const unsigned G1 = (unsigned)((n + T - 1) / T), G2 = (unsigned)((n / 2 + T - 1) / T);
    Ptrs q3 = p; for (int z = 0; z < 12; z++) q3.r[z] = p.r[z] + 3; for (int z = 0; z < 8; z++) q3.w[z] = p.w[z] + 3;     // 24 B into the line
#define FLAT(NR, NW) \
    run("flat  8 B/lane aligned", NR, NW, (double)n, [&] { rw1<NR, NW><<<G1, T>>>(p, n); }); \
    run("flat 16 B/lane aligned", NR, NW, (double)n, [&] { rw2<NR, NW><<<G2, T>>>(p, n); }); \
    run("flat  8 B/lane +24 B", NR, NW, (double)n, [&] { rw1<NR, NW><<<G1, T>>>(q3, n); }); \
    run("flat 16 B/lane +24 B", NR, NW, (double)n, [&] { rw2<NR, NW><<<G2, T>>>(q3, n); });
    FLAT(1, 1) FLAT(7, 4) FLAT(1, 3) FLAT(10, 6) FLAT(4, 3)
    // tile march
    const dim3 g(nx / 64, ny / 8), b(64, 8);
    const double cells = (double)nx * ny * nz;
#define MARCH(NR, NW) { \
    const size_t l = sizeof(double) * 2 * NR * TN; \
    CK(hipFuncSetAttribute((const void *)tile_march<NR, NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l)); \
    CK(hipFuncSetAttribute((const void *)tile_march<NR, NW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l)); \
    run("tile march, register staging", NR, NW, cells, [&] { tile_march<NR, NW, 0><<<g, b, l>>>(p, nx, ny, nz, pitch); }); \
    CK(hipMemcpy(buf[19], buf[12], n * 8, hipMemcpyDeviceToDevice)); \
    run("tile march, LDS-DMA", NR, NW, cells, [&] { tile_march<NR, NW, 1><<<g, b, l>>>(p, nx, ny, nz, pitch); }); \
    if (NW > 0) { unsigned long long *cnt, h = 0; CK(hipMalloc((void **)&cnt, 8)); CK(hipMemset(cnt, 0, 8)); \
      diff<<<(n + 255) / 256, 256>>>(buf[19], buf[12], n, cnt); CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost)); \
      printf("    LDS-DMA result differs from register staging in %llu elements\n", h); CK(hipFree(cnt)); } }
    MARCH(4, 2) MARCH(8, 2) MARCH(6, 1) MARCH(10, 3)
    CK(hipGetLastError());
    return 0;
}
