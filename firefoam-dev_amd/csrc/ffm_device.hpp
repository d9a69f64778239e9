// ffm_device.hpp -- device-side helpers shared by the kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>

// wave64 butterfly-free, order-fixed reductions: lane 0 ends with the result.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

// Block reductions (blockDim.x a multiple of 64, <= 1024).  Result valid in
// thread 0.  sm must hold blockDim.x/64 doubles.  Waves are added in wave order.
__device__ __forceinline__ double block_sum(double v, double *sm)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int i = 0; i < nw; i++) r += sm[i];
    }
    return r;
}
__device__ __forceinline__ double block_min(double v, double *sm)
{
    v = wave_min(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = v;
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        r = sm[0];
        for (int i = 1; i < nw; i++) r = fmin(r, sm[i]);
    }
    return r;
}
__device__ __forceinline__ double block_max(double v, double *sm)
{
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = v;
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        r = sm[0];
        for (int i = 1; i < nw; i++) r = fmax(r, sm[i]);
    }
    return r;
}

// ------------------------------------------------------ sliced owner-ELL view ---
// Passed by value to every kernel that walks matrix rows (layout: ffm_internal.hpp).
struct LduView {
    int N;                      // number of ROWS (owned cells); column indices may reach the ghost cells beyond
    const int *upOff, *loOff;   // [nSlices+1]
    const int *upNbr;           // [upTotal]
    const int *loEnt;           // [loTotal]
    int upW;                    // uniform upper width of every slice, or -1
    int loW;                    // uniform lower width of every slice, or -1
    const int *sched; int nSched;   // XCD-aware chunk schedule of the row kernels (256 rows per chunk)
};

// first native face index of the slice of cell c
__device__ __forceinline__ int up_base(const LduView &v, int sl) { return v.upW >= 0 ? sl * v.upW * 64 : v.upOff[sl]; }
__device__ __forceinline__ int up_width(const LduView &v, int sl) { return v.upW >= 0 ? v.upW : (v.upOff[sl + 1] - v.upOff[sl]) >> 6; }
__device__ __forceinline__ int lo_base(const LduView &v, int sl) { return v.loW >= 0 ? sl * v.loW * 64 : v.loOff[sl]; }
__device__ __forceinline__ int lo_width(const LduView &v, int sl) { return v.loW >= 0 ? v.loW : (v.loOff[sl + 1] - v.loOff[sl]) >> 6; }
// native face index of slot `slot` in the row of cell `cell`
__device__ __forceinline__ int face_of(const LduView &v, int cell, int slot) { return up_base(v, cell >> 6) + slot * 64 + (cell & 63); }

// Row loaders: fetch all lower / upper entries of the row of cell c into registers with
// predicated, fully unrolled loads (W = compile-time bound on the slot count), so that the
// index loads of all slots are in flight together and the dependent gathers likewise: a row
// costs two memory round trips instead of two per slot.  Padding slots point at harmless
// addresses (own cell, face 0) and are masked out of the arithmetic.
template <int W> struct RowEnt { int nb[W]; int f[W]; bool on[W]; };

// NT: index streams are read once per kernel -- load them non-temporally so they do not evict the gather targets
// (x, coefficients) from the XCD's L2
template <int W, bool NT = false>
__device__ __forceinline__ void load_lower(const LduView &v, int c, RowEnt<W> &R)
{
    const int sl = c >> 6, lane = c & 63;
    const int lb = lo_base(v, sl), lw = lo_width(v, sl);
    int e[W];
#pragma unroll
    for (int s = 0; s < W; s++) e[s] = (s < lw) ? (NT ? __builtin_nontemporal_load(&v.loEnt[lb + s * 64 + lane]) : v.loEnt[lb + s * 64 + lane]) : -1;
#pragma unroll
    for (int s = 0; s < W; s++) {
        R.on[s] = e[s] >= 0;
        R.nb[s] = R.on[s] ? (e[s] >> 4) : c;
        R.f[s] = R.on[s] ? face_of(v, R.nb[s], e[s] & 15) : 0;
    }
}

// MASK_GHOST: drop upper neighbours that are ghost cells (index >= v.N): block-Jacobi sweeps
template <int W, bool MASK_GHOST = false, bool NT = false>
__device__ __forceinline__ void load_upper(const LduView &v, int c, RowEnt<W> &R)
{
    const int sl = c >> 6, lane = c & 63;
    const int ub = up_base(v, sl), uw = up_width(v, sl);
#pragma unroll
    for (int s = 0; s < W; s++) {
        const int idx = ub + s * 64 + lane;
        const int n = (s < uw) ? (NT ? __builtin_nontemporal_load(&v.upNbr[idx]) : v.upNbr[idx]) : -1;
        R.on[s] = n >= 0 && (!MASK_GHOST || n < v.N);
        R.nb[s] = R.on[s] ? n : c;
        R.f[s] = R.on[s] ? idx : 0;
    }
}

// (W = 3: hexahedral meshes -- at most 3 lower and 3 upper neighbours per cell -- get their own instantiation: a quarter fewer
// predicated slot loads and registers than W = 4 in every row kernel)
#define FFM_DISPATCH_W(maxW, CALL)                      \
    do {                                                \
        if ((maxW) <= 3) { constexpr int W = 3; CALL; } \
        else if ((maxW) <= 4) { constexpr int W = 4; CALL; } \
        else if ((maxW) <= 8) { constexpr int W = 8; CALL; } \
        else if ((maxW) <= 16) { constexpr int W = 16; CALL; } \
        else { constexpr int W = 32; CALL; }            \
    } while (0)
