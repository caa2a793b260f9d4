// block-level reductions shared by the kernels of this translation unit.
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// (WAVE, dpp_move, wave_sum: rowdev.hpp - shared with the long-grid kernels of longgrid.hip)

// Sum over a 1-D block (blockDim.x multiple of 64, <= 1024).  Result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* lds /* >= 16 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    const int nw = (blockDim.x + WAVE - 1) >> 6;
    double r = 0.0;
    if (w == 0) {
        r = (lane < nw) ? lds[lane] : 0.0;
        r = wave_sum(r);
    }
    return r;
}

// Sum N per-thread values over a 1-D block with two barriers in total (instead of 2N):
// wave shuffles, one LDS exchange of N x (#waves) partials, fixed-order final sum.  The
// results are valid in threads 0..N-1 (thread i holds the total of v[i]).  lds >= N*16 doubles.
template <int N>
__device__ __forceinline__ double block_sum_multi(const double (&v)[N], double* lds) {
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    const int nw = (blockDim.x + WAVE - 1) >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0) lds[i * 16 + w] = s;
    }
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x < N)
        for (int k = 0; k < nw; ++k) r += lds[threadIdx.x * 16 + k];
    return r;
}

}  // namespace hmg
