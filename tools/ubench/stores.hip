// Why does k_stretch (3 stored arrays, two 2-D loads per column) store at 3.7 TB/s when k_vmix (4 stored arrays, no loads) stores at 6.2?
// Variants of a pure-store kernel over dense 2054 x 2054 x 50 arrays (tools only; not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct P { double *w[4]; const double *h, *zt; const double *tab; };
// one element per thread, linear over the whole array; NW arrays; LOADS: value from two 2-D loads; KC levels per thread
template <int NW, int LOADS, int KC> __global__ void __launch_bounds__(256) st(P p, int nxy, int nz)
{
    const int c = blockIdx.x * 256 + threadIdx.x;           // column (linear in the plane)
    if (c >= nxy) return;
    const int k0 = blockIdx.y * KC;
    double a = 1.0, b = 2.0;
    if (LOADS) { a = p.h[c]; b = p.zt[c]; }
    const double hinv = 1.0 / (3.0 + a);
#pragma unroll
    for (int q = 0; q < KC; q++) {
        const int k = k0 + q;
        if (k >= nz) break;
        const double t = LOADS == 2 ? p.tab[k] : (double)k;
        const double v = b + (b + a) * ((t + 0.5 * a) * hinv);
#pragma unroll
        for (int n = 0; n < NW; n++) p.w[n][(size_t)k * nxy + c] = v + (double)n;
    }
}
int main()
{
    const int nx = 2054, ny = 2054, nz = 50, nxy = nx * ny;
    const size_t n = (size_t)nxy * nz;
    P p; for (int q = 0; q < 4; q++) CK(hipMalloc((void **)&p.w[q], n * 8));
    double *h, *zt, *tab; CK(hipMalloc((void **)&h, (size_t)nxy * 8)); CK(hipMalloc((void **)&zt, (size_t)nxy * 8)); CK(hipMalloc((void **)&tab, 64 * 8));
    CK(hipMemset(h, 0, (size_t)nxy * 8)); CK(hipMemset(zt, 0, (size_t)nxy * 8)); CK(hipMemset(tab, 0, 64 * 8));
    p.h = h; p.zt = zt; p.tab = tab;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int nw, auto launch) {
        launch(); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-52s %d w  %8.3f ms  %6.2f TB/s\n", name, nw, best, 8.0 * (double)n * nw / (best * 1e-3) / 1e12); fflush(stdout);
    };
    const unsigned gx = (nxy + 255) / 256;
#define V(NW, LOADS, KC, name) run(name, NW, [&] { st<NW, LOADS, KC><<<dim3(gx, (nz + KC - 1) / KC), 256>>>(p, nxy, nz); });
    V(4, 0, 1, "4 arrays, no loads, 1 level per thread (k_vmix)")
    V(3, 0, 1, "3 arrays, no loads, 1 level per thread")
    V(3, 1, 1, "3 arrays, two 2-D loads, 1 level per thread")
    V(3, 2, 1, "3 arrays, two 2-D loads + table, 1 level per thread")
    V(3, 0, 8, "3 arrays, no loads, 8 levels per thread")
    V(3, 1, 8, "3 arrays, two 2-D loads, 8 levels per thread")
    V(3, 1, 2, "3 arrays, two 2-D loads, 2 levels per thread")
    V(3, 1, 4, "3 arrays, two 2-D loads, 4 levels per thread")
    V(3, 1, 50, "3 arrays, two 2-D loads, 50 levels per thread")
    V(1, 0, 1, "1 array, no loads, 1 level per thread")
    V(1, 1, 1, "1 array, two 2-D loads, 1 level per thread")
    CK(hipGetLastError());
    return 0;
}
