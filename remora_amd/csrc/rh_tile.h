// rh_tile.h -- LDS-staged 2-D tiles with halo for the horizontal stencils.
//
// Measured on MI355X (profiles/): a neighbour re-load (i+-1, i+-2, j+-1, ...) issued while the line is still in flight is not
// merged by L1/L2, so a kernel that reads its stencil straight from global memory moves ~(load instructions x 8 B) over the
// fabric: 2-7x the algorithmic bytes.  Here every block stages one (64+2HX) x (BY+2HY) tile per field and level in LDS: each
// thread contributes the value of its own point (which it loads exactly once), the first threads of the block fetch the
// halo ring, and all stencil neighbours are then read from LDS.  Tiles are double-buffered on the parity of k so that a
// k-marching kernel needs ONE __syncthreads() per level.
#pragma once
#include "rh_common.h"

#define RT_BX 64          // threads along i: one wavefront, 512-B rows
#define RT_BY 8           // rows of j per block (8 wavefronts)
static inline dim3 rt_block() { return dim3(RT_BX, RT_BY, 1); }
static inline dim3 rt_grid(const rh_box &b) { return dim3((b.hi[0] - b.lo[0] + RT_BX) / RT_BX, (b.hi[1] - b.lo[1] + RT_BY) / RT_BY, 1); }

template <int HX, int HY, int BY = RT_BY>
struct Tile {
    static constexpr int PX = RT_BX + 2 * HX, PY = BY + 2 * HY, SIZE = PX * PY;
    static constexpr int NROWCELLS = 2 * HY * PX, NHALO = NROWCELLS + 2 * HX * BY;
    double *s;
    // value at tile-relative offset (di,dj) from this thread's own point
    __device__ __forceinline__ double at(int tx, int ty, int di, int dj) const { return s[(ty + HY + dj) * PX + (tx + HX + di)]; }
    // stage level k of `a`: own point value `c` (already loaded by the caller), halo ring fetched here.  (i0,j0) = global index
    // of thread (0,0)'s point.  Cells outside the array are left untouched (no consumer inside the kernel's box reads them).
    __device__ __forceinline__ void fill(const A4 &a, int k, int i0, int j0, int tx, int ty, double c) const
    {
        s[(ty + HY) * PX + (tx + HX)] = c;
        if (NHALO > 0) {
            for (int t = ty * RT_BX + tx; t < NHALO; t += RT_BX * BY) {
                int x, y;
                if (t < NROWCELLS) { const int r = t / PX; x = t - r * PX; y = r < HY ? r : BY + r; }
                else { const int u = t - NROWCELLS; constexpr int W2 = 2 * HX > 0 ? 2 * HX : 1; const int r = u / W2; const int xx = u - r * W2; y = HY + r; x = xx < HX ? xx : RT_BX + xx; }
                const int gi = i0 - HX + x, gj = j0 - HY + y;
                if (gi >= a.lo0 && gi <= a.hi0 && gj >= a.lo1 && gj <= a.hi1) s[y * PX + x] = a(gi, gj, k);
            }
        }
    }
};
// array-like view of a staged tile: v(i,j,k) with GLOBAL indices (k ignored), so stencil code written for A4 runs unchanged
template <int HX, int HY, int BY = RT_BY>
struct TView {
    const double *s; int i0, j0;
    __device__ __forceinline__ double operator()(int i, int j, int) const { return s[(j - j0 + HY) * (RT_BX + 2 * HX) + (i - i0 + HX)]; }
};
// Software-pipelined form: a thread's halo cell is fixed for the whole kernel, so it is located once (HaloCell), its value
// for level k+1 is fetched into a register while level k is being computed, and put() writes own value + halo value into LDS
// at the top of the next iteration -- no load sits between the LDS stores and the barrier.  One halo cell per thread at most.
struct HaloCell { int lds; unsigned g; };    // LDS offset inside the tile (-1: none) and element offset inside one k-plane of the array
template <int HX, int HY, int BY = RT_BY>
__device__ __forceinline__ HaloCell halo_cell(const A4 &a, int i0, int j0, int tx, int ty)
{
    typedef Tile<HX, HY, BY> T;
    static_assert(T::NHALO <= RT_BX * BY, "one halo cell per thread");
    HaloCell h; h.lds = -1; h.g = 0;
    const int t = ty * RT_BX + tx;
    if (t < T::NHALO) {
        int x, y;
        if (t < T::NROWCELLS) { const int r = t / T::PX; x = t - r * T::PX; y = r < HY ? r : BY + r; }
        else { constexpr int W2 = 2 * HX > 0 ? 2 * HX : 1; const int u = t - T::NROWCELLS; const int r = u / W2; const int xx = u - r * W2; y = HY + r; x = xx < HX ? xx : RT_BX + xx; }
        const int gi = i0 - HX + x, gj = j0 - HY + y;
        if (gi >= a.lo0 && gi <= a.hi0 && gj >= a.lo1 && gj <= a.hi1) { h.lds = y * T::PX + x; h.g = (unsigned)(gi - a.lo0) + (unsigned)(gj - a.lo1) * a.sy; }
    }
    return h;
}
__device__ __forceinline__ double halo_ld(const A4 &a, const HaloCell &h, int k) { return h.lds >= 0 ? a.p[h.g + (unsigned)(k - a.lo2) * a.sz] : 0.0; }
template <int HX, int HY>
__device__ __forceinline__ void tile_put(double *s, int tx, int ty, double own, const HaloCell &h, double hv)
{
    s[(ty + HY) * Tile<HX, HY>::PX + (tx + HX)] = own;
    if (h.lds >= 0) s[h.lds] = hv;
}
// Column cursors.  In a k-march the (i, j) part of every own-point and halo-cell access is the same at every level, but written
// as a(i, j, k) each access needs the array's whole descriptor (12 scalar registers per array): with 8-20 arrays in a kernel the
// scalar file overflows and the compiler parks descriptors in vector lanes, re-reading them lane by lane inside the loop
// (k_rhs_uv_tiled: 475 v_readlane / v_writelane before).  A cursor keeps, per thread, the address of the column's level 0 (null
// if the point lies outside the array) and the level stride: a level is one multiply-add away and the descriptor dies after the
// prologue.
struct ColPtr { double *p; unsigned sz; };
__device__ __forceinline__ ColPtr col_ptr(const A4 &a, int i, int j, int n = 0)
{
    ColPtr c; c.sz = a.sz;
    c.p = (i >= a.lo0 && i <= a.hi0 && j >= a.lo1 && j <= a.hi1) ? a.p + ((long)a.ix(i, j, a.lo2) + (long)n * a.sn - (long)a.lo2 * a.sz) : nullptr;
    return c;
}
__device__ __forceinline__ ColPtr halo_ptr(const A4 &a, const HaloCell &h)
{
    ColPtr c; c.sz = a.sz; c.p = h.lds >= 0 ? a.p + ((long)h.g - (long)a.lo2 * a.sz) : nullptr;
    return c;
}
__device__ __forceinline__ double col_ld(const ColPtr &c, int k) { return c.p ? c.p[(unsigned)k * c.sz] : 0.0; }
// Cursors that can always be dereferenced: a point outside the array (or a thread without a halo cell) gets the array's first element
// with stride 0, whose value the caller never uses.  A k-march can then issue every load of a level unconditionally -- no exec-mask
// branch around each one (k_rhs_uv_tiled: 14 of them per level), and the loads of a level leave as one clause.
__device__ __forceinline__ ColPtr col_ptr_any(const A4 &a, int i, int j, int n = 0)
{
    ColPtr c = col_ptr(a, i, j, n);
    if (!c.p) { c.p = a.p; c.sz = 0; }
    return c;
}
__device__ __forceinline__ ColPtr halo_ptr_any(const A4 &a, const HaloCell &h)
{
    ColPtr c = halo_ptr(a, h);
    if (!c.p) { c.p = a.p; c.sz = 0; }
    return c;
}
__device__ __forceinline__ double col_ldu(const ColPtr &c, int k) { return c.p[(unsigned)k * c.sz]; }
// The same with the level stride left to the array (one scalar for all cursors of an array instead of a vector register per cursor): the
// stand-in for a point outside the array is the array's first column, every level of which exists.
__device__ __forceinline__ double *col_base_any(const A4 &a, int i, int j, int n = 0)
{
    const bool in = i >= a.lo0 && i <= a.hi0 && j >= a.lo1 && j <= a.hi1;
    return a.p + ((long)a.ix(in ? i : a.lo0, in ? j : a.lo1, a.lo2) + (long)n * a.sn - (long)a.lo2 * a.sz);
}
__device__ __forceinline__ double *halo_base_any(const A4 &a, const HaloCell &h) { return a.p + ((h.lds >= 0 ? (long)h.g : 0L) - (long)a.lo2 * a.sz); }
__device__ __forceinline__ void col_st(const ColPtr &c, int k, double v) { c.p[(unsigned)k * c.sz] = v; }

// The same for kernels whose threads only ever touch their own column: a view with the call syntax of A4 whose (i, j) arguments are
// ignored -- a(i, j, k[, n]) is level k (component n) of THE column the view was made for.  The point must lie inside the array.
struct ColV {
    double *p; unsigned sz, sn;
    __device__ __forceinline__ double &operator()(int, int, int k) const { return p[(unsigned)k * sz]; }
    __device__ __forceinline__ double &operator()(int, int, int k, int n) const { return p[(unsigned)k * sz + (unsigned)n * sn]; }
};
__device__ __forceinline__ ColV col_view(const A4 &a, int i, int j)
{ ColV c; c.sz = a.sz; c.sn = a.sn; c.p = a.p + ((long)a.ix(i, j, a.lo2) - (long)a.lo2 * a.sz); return c; }

// own-point load guarded by the array box (threads of a partial block outside the array contribute nothing)
__device__ __forceinline__ double ld_in(const A4 &a, int i, int j, int k)
{ return (i >= a.lo0 && i <= a.hi0 && j >= a.lo1 && j <= a.hi1) ? a(i, j, k) : 0.0; }
