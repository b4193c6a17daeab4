// Small device helpers and reduction kernels shared by the translation units of
// libfsamd.so (each TU gets its own static copy; no relocatable device code needed).
#pragma once
#include "fs_common.h"

// ---- XCD-aware persistent chunk mapping -------------------------------------------------
// Workgroup b is (observed) placed on XCD b % 8.  Give every XCD one contiguous 1/8 of the
// chunk range so the x-vector lines gathered by neighbouring rows stay in that XCD's L2.
// Correctness never depends on the placement.
struct chunk_iter {
    int64_t cur, end, step;
};
__device__ __forceinline__ chunk_iter xcd_chunks(int64_t n_chunks) {
    const int64_t per_xcd = (n_chunks + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int64_t j = blockIdx.x >> 3;
    chunk_iter it;
    it.step = gridDim.x >> 3;
    it.cur = xcd * per_xcd + j;
    const int64_t e = (xcd + 1) * per_xcd;
    it.end = e < n_chunks ? e : n_chunks;
    return it;
}

// wave64 shuffle reduction -> one LDS slot per wave -> thread 0 holds the block sum.
// Fixed order, so a given launch geometry always produces the same bits.
__device__ __forceinline__ double fs_block_sum(double v, double* lds4) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds4[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) t = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    __syncthreads();
    return t;
}

static __global__ void __launch_bounds__(FS_BLOCK) k_dot_partial(const double* __restrict__ x,
                                                                 const double* __restrict__ y, int64_t n,
                                                                 double* __restrict__ partial) {
    __shared__ double lds4[4];
    double acc = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) acc += x[i] * y[i];
    const double t = fs_block_sum(acc, lds4);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// fixed-order sum of nsums (<= 4) interleaved partial arrays (partial[j*count + i]) by ONE workgroup of
// 1024 threads -> out[j].  Latency-bound (it sits between the SpMV and the all-reduce on N>1 GPUs), so
// all loads of a thread are issued before the first reduction step.
#define FS_SUM_BLOCK 1024
static __global__ void __launch_bounds__(FS_SUM_BLOCK) k_sum_partials(const double* __restrict__ partial, int count,
                                                                      int nsums, double* __restrict__ out) {
    __shared__ double lds[16][4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < count; i += FS_SUM_BLOCK) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nsums) acc[j] += partial[(int64_t)j * count + i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[wave][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 4 && (int)threadIdx.x < nsums) {
        double t = 0.0;
        for (int w = 0; w < FS_SUM_BLOCK / 64; ++w) t += lds[w][threadIdx.x];   // fixed order
        out[threadIdx.x] = t;
    }
}
