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
