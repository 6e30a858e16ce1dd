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
        printf("%-28s %2d r + %d w  %8.3f ms  %6.2f TB/s\n", name, nr, nw, best, 8.0 * n * (nr + nw) / (best * 1e-3) / 1e12); fflush(stdout);
    };
    run("copy", 1, 1, [&] { rw<1, 1><<<G, T>>>(p, n); });
    run("massflux_omega-like", 4, 3, [&] { rw<4, 3><<<G, T>>>(p, n); });
    run("uv3dmix-like", 5, 2, [&] { rw<5, 2><<<G, T>>>(p, n); });
    run("prestep_uv-like", 7, 4, [&] { rw<7, 4><<<G, T>>>(p, n); });
    run("rhs_uv_3d-like", 8, 2, [&] { rw<8, 2><<<G, T>>>(p, n); });
    run("rhs_t3-like", 10, 3, [&] { rw<10, 3><<<G, T>>>(p, n); });
    run("prestep_t3-like", 10, 6, [&] { rw<10, 6><<<G, T>>>(p, n); });
    run("12 r + 8 w", 12, 8, [&] { rw<12, 8><<<G, T>>>(p, n); });
    return 0;
}
