// Streaming-bandwidth probe for the access shapes the path's kernels use (tools only; not part of the library).
// copy: d[i] = s[i];  triad-like: d[i] = a[i] + b[i] * c[i]  over FP64 arrays far larger than the Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));     // 16-B access that is only 8-B aligned
typedef double d2a __attribute__((ext_vector_type(2)));

__global__ void c1(double *d, const double *s, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = s[i]; }
template <int U> __global__ void cU(double *d, const double *s, long n)
{   // each block handles U consecutive 256-element rows, all loads first
    long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x; double v[U];
    #pragma unroll
    for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; v[u] = i < n ? s[i] : 0.0; }
    #pragma unroll
    for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n) d[i] = v[u]; }
}
template <class V, int U> __global__ void cV(double *d, const double *s, long n2)     // n2 = number of 2-element vectors
{
    long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x; V v[U];
    const V *sv = (const V *)s; V *dv = (V *)d;
    #pragma unroll
    for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n2) v[u] = sv[i]; }
    #pragma unroll
    for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n2) dv[i] = v[u]; }
}
template <int U> __global__ void t3(double *d, const double *a, const double *b, const double *c, long n)
{
    long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x; double x[U], y[U], z[U];
    #pragma unroll
    for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n) { x[u] = a[i]; y[u] = b[i]; z[u] = c[i]; } }
    #pragma unroll
    for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n) d[i] = x[u] + y[u] * z[u]; }
}
template <class V, int U> __global__ void t3V(double *d, const double *a, const double *b, const double *c, long n2)
{
    long base = (long)blockIdx.x * blockDim.x * U + threadIdx.x; V x[U], y[U], z[U];
    #pragma unroll
    for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n2) { x[u] = ((const V *)a)[i]; y[u] = ((const V *)b)[i]; z[u] = ((const V *)c)[i]; } }
    #pragma unroll
    for (int u = 0; u < U; u++) { long i = base + (long)u * blockDim.x; if (i < n2) ((V *)d)[i] = x[u] + y[u] * z[u]; }
}
__global__ void r1(double *d, const double *s, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { double v = s[i]; if (v == 1.2345e300) d[i] = v; } }
__global__ void w1(double *d, long n, double v) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = v; }
__global__ void r2w1(double *d, const double *a, const double *b, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) d[i] = a[i] + b[i]; }
__global__ void r2w2(double *d, double *e, const double *a, const double *b, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { double x = a[i], y = b[i]; d[i] = x + y; e[i] = x - y; } }
__global__ void fill_rand(double *d, long n, unsigned long long seed) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { unsigned long long x = (unsigned long long)i * 6364136223846793005ULL + seed; x ^= x >> 29; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 32; d[i] = 1.0 + (double)(x >> 11) * (1.0 / 9007199254740992.0); } }
int main(int argc, char **argv)
{
    const long n = 2054L * 2054 * 50;           // one 3-D component of the bench grid (1.69 GB)
    double *buf[4]; for (auto &p : buf) { CK(hipMalloc((void **)&p, (n + 2) * 8)); CK(hipMemset(p, 0, (n + 2) * 8)); }
    if (argc > 1) { for (int q = 0; q < 4; q++) fill_rand<<<(n + 255) / 256, 256>>>(buf[q], n, 12345ULL + q); hipDeviceSynchronize(); printf("buffers hold random doubles in [1, 2)\n"); }
    else printf("buffers hold zeros\n");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, double bytes, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-34s %8.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12); fflush(stdout);
    };
    const int T = 256; const long n2 = n / 2;
    run("read only 8B/lane, 1 elem/thread", 8.0 * n, [&] { r1<<<(n + T - 1) / T, T>>>(buf[0], buf[1], n); });
    run("write only 8B/lane, 1 elem/thread", 8.0 * n, [&] { w1<<<(n + T - 1) / T, T>>>(buf[0], n, 1.0); });
    run("2r1w 8B/lane, 1 elem/thread", 24.0 * n, [&] { r2w1<<<(n + T - 1) / T, T>>>(buf[0], buf[1], buf[2], n); });
    run("2r2w 8B/lane, 1 elem/thread", 32.0 * n, [&] { r2w2<<<(n + T - 1) / T, T>>>(buf[0], buf[3], buf[1], buf[2], n); });
    run("copy 8B/lane, 1 elem/thread", 16.0 * n, [&] { c1<<<(n + T - 1) / T, T>>>(buf[0], buf[1], n); });
    run("copy 8B/lane, 4 elems/thread", 16.0 * n, [&] { cU<4><<<(n + T * 4 - 1) / (T * 4), T>>>(buf[0], buf[1], n); });
    run("copy 8B/lane, 8 elems/thread", 16.0 * n, [&] { cU<8><<<(n + T * 8 - 1) / (T * 8), T>>>(buf[0], buf[1], n); });
    run("copy 16B/lane aligned, 1/thread", 16.0 * n, [&] { cV<d2a, 1><<<(n2 + T - 1) / T, T>>>(buf[0], buf[1], n2); });
    run("copy 16B/lane aligned, 4/thread", 16.0 * n, [&] { cV<d2a, 4><<<(n2 + T * 4 - 1) / (T * 4), T>>>(buf[0], buf[1], n2); });
    run("copy 16B/lane +8B offset, 4/thread", 16.0 * n, [&] { cV<d2u, 4><<<(n2 + T * 4 - 1) / (T * 4), T>>>(buf[0] + 1, buf[1] + 1, n2); });
    run("3r1w 8B/lane, 1/thread", 32.0 * n, [&] { t3<1><<<(n + T - 1) / T, T>>>(buf[0], buf[1], buf[2], buf[3], n); });
    run("3r1w 8B/lane, 4/thread", 32.0 * n, [&] { t3<4><<<(n + T * 4 - 1) / (T * 4), T>>>(buf[0], buf[1], buf[2], buf[3], n); });
    run("3r1w 16B/lane aligned, 2/thread", 32.0 * n, [&] { t3V<d2a, 2><<<(n2 + T * 2 - 1) / (T * 2), T>>>(buf[0], buf[1], buf[2], buf[3], n2); });
    run("3r1w 16B/lane +8B offset, 2/thread", 32.0 * n, [&] { t3V<d2u, 2><<<(n2 + T * 2 - 1) / (T * 2), T>>>(buf[0] + 1, buf[1] + 1, buf[2] + 1, buf[3] + 1, n2); });
    return 0;
}
