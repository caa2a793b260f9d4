// K1 - sigma^2(z,m): the fp64-MFMA contraction over k' and its ordered second stage (hmvec/cosmology.py:245-269).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- K1: sigma^2(z,m) (A2)
// sigma2[z,m] = sum_j P[z,j] * A[j,m],  A[j,m] = wq[j] W(kq[j] R[m])^2, is the one dense
// contraction of the path (nz x nm x 10^4): it runs on the fp64 matrix cores.  A wavefront
// owns a 16-mass tile, 16*ZB redshifts and one segment of the k' axis; per step of four k'
// every lane evaluates ONE window value - which is directly its element of the MFMA B operand
// (B[k = lane>>4][col = lane&15]) - loads its element(s) of P from a [k'][z] transposed,
// zero-padded copy (A[row = lane&15][k = lane>>4], 128 B coalesced per 16 lanes), and issues
// ZB v_mfma_f64_16x16x4_f64.  The window (one sincos) is therefore evaluated exactly once per
// (m, k') for up to 32 redshifts, and the nz-fold multiply-accumulate is off the vector ALU.
// The k' axis is cut into a number of segments that depends on nq only, so the summation
// order - and the result, bit for bit - is the same for a z-slab and for the full grid; the
// per-segment partial sums are combined in order by sigma2_combine_kernel.  Nothing of shape
// (nz,nm,nq) is materialised (the reference builds 1.3 GB temporaries here).
typedef double d4_t __attribute__((ext_vector_type(4)));
#ifndef HMG_SIG_SEG_LEN
#define HMG_SIG_SEG_LEN 80
#endif
constexpr int SIG_SEG_LEN = HMG_SIG_SEG_LEN;    // k' values per segment (multiple of 16)

// out[c][r] = in[r][c] for r < rows, zero for rows <= r < rows_pad
__global__ void transpose_pad_kernel(int rows, int rows_pad, int cols, const double* __restrict__ in,
                                     double* __restrict__ out /*[cols][rows_pad]*/) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows_pad * cols) return;
    const int c = (int)(i / rows_pad), r = (int)(i - (size_t)c * rows_pad);
    out[i] = r < rows ? in[(size_t)r * cols + c] : 0.0;
}

// (132 VGPRs, 3 waves/SIMD; forcing 4 spills and is no faster: 23.5 vs 24.0 us at Config 3)
#ifndef HMG_SIG_OCC
#define HMG_SIG_OCC 1
#endif
template <int ZB>
__device__ __forceinline__ void sigma2_mfma_block(int bx, int seg, int bz, int nz, int nzp, int nm, int nq,
                                                  const double* __restrict__ PT /*[nq][nzp]*/,
                                                  const double* __restrict__ kq,
                                                  const double* __restrict__ wq,
                                                  const double* __restrict__ R, double tswitch,
                                                  double* __restrict__ partial /*[seg][nz][nm]*/) {
    const int lane = threadIdx.x & 63, col = lane & 15, kk = lane >> 4;
    const int m = bx * 16 + col;
    const int z0 = bz * (16 * ZB);
    const double r = R[min(m, nm - 1)];
    d4_t acc[ZB];
#pragma unroll
    for (int b = 0; b < ZB; ++b) acc[b] = d4_t{0.0, 0.0, 0.0, 0.0};
    const int q_lo = seg * SIG_SEG_LEN, q_hi = min(nq, q_lo + SIG_SEG_LEN);
    constexpr int NT = SIG_SEG_LEN / 16;      // trips of four MFMA k-steps
    // Two-stage pipeline over the trips: the loads of trip t+1 (k', quadrature weight, the P rows; positions
    // past the end of the segment are clamped and given zero weight) are issued before trip t's window
    // values are evaluated, so a wavefront holds two trips of operands instead of the whole segment
    // (236 -> ~120 VGPRs: four wavefronts per SIMD instead of two, which is what feeds the VALU here).
    // The MFMA accumulation order over k' is unchanged.
    struct Trip { double kv[4], wv[4], pv[4][ZB]; };
    auto load_trip = [&](Trip& T, int t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q_lo + 16 * t + 4 * u + kk;
            const int qc = min(q, nq - 1);
            T.kv[u] = kq[qc];
            const double w = wq[qc];
            T.wv[u] = (q < q_hi) ? w : 0.0;
            const double* __restrict__ prow = PT + (size_t)qc * nzp + z0 + col;
#pragma unroll
            for (int b = 0; b < ZB; ++b) T.pv[u][b] = prow[16 * b];
        }
    };
    // window values (branch-free: Taylor and trigonometric forms both evaluated, selected by kR; the library
    // sincos is only called if some lane has kR >= 1e9) and the MFMA accumulation.  (Round 3: skipping the
    // trigonometric form on trips whose 64 values all lie below the Taylor switch - a quarter of the trips of a
    // default grid - was measured and is slower, 19.2 -> 20.4 us at nz = 4, 28.7 -> 29.6 at nz = 32: the branch
    // splits the four interleaved evaluations the scheduler overlaps.)
    auto consume = [&](const Trip& T) {
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double kR = T.kv[u] * r;
            const double xx = kR * kR;
            const double wt = 1.0 - 0.1 * xx + 0.00357142857143 * xx * xx;
            // numerator sin kR - kR cos kR up to its sign, which the square below does not see (sici.hpp)
            double num = sin_minus_ycos_nosign(fmin(kR, 1.0e9), kR);
#ifndef HMG_SIG_NOSLOW
            if (__builtin_expect(__any(kR >= 1.0e9), 0)) {
                if (kR >= 1.0e9) {
                    double sn, cs;
                    sincos(kR, &sn, &cs);
                    num = sn - kR * cs;
                }
            }
#endif
            const double wtr = 3.0 * num * rcp_fast(fmax(xx * kR, 1.0e-300));
            const double w = (kR < tswitch) ? wt : wtr;
            a[u] = T.wv[u] * (w * w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < ZB; ++b)
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(T.pv[u][b], a[u], acc[b], 0, 0, 0);
    };
    Trip ta, tb;
    load_trip(ta, 0);
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
        if (t + 1 < NT) load_trip(tb, t + 1);
        consume(ta);
        if (t + 2 < NT) load_trip(ta, t + 2);
        if (t + 1 < NT) consume(tb);
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    if (m < nm) {
#pragma unroll
        for (int b = 0; b < ZB; ++b)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int z = z0 + 16 * b + kk + 4 * rg;
                if (z < nz) partial[((size_t)seg * nz + z) * nm + m] = acc[b][rg];
            }
    }
}
template <int ZB>
__global__ __launch_bounds__(64, HMG_SIG_OCC) void sigma2_mfma_kernel(int nz, int nzp, int nm, int nq,
                                                         const double* __restrict__ PT, const double* __restrict__ kq,
                                                         const double* __restrict__ wq, const double* __restrict__ R,
                                                         double tswitch, double* __restrict__ partial) {
    sigma2_mfma_block<ZB>(blockIdx.x, blockIdx.y, blockIdx.z, nz, nzp, nm, nq, PT, kq, wq, R, tswitch, partial);
}

__device__ __forceinline__ double sigma2_segment_sum(int n, int parts, const double* __restrict__ partial, size_t i, int w) {
    // parts w, w+4, w+8, ... in order.  Sixteen loads are in flight per round (they do not depend on the
    // running sum); positions past the end contribute +0.0, which leaves the sum's bits alone - a scalar
    // tail loop here cost one memory round trip per leftover part (7 of them at nq = 10^4).
    double s = 0.0;
    for (int p = w; p < parts; p += 64) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int pp = p + 4 * u;
            v[u] = pp < parts ? partial[(size_t)pp * n + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    return s;
}
// out[i] = sum_p partial[p][i] in a fixed order: 4 wavefronts per 64 outputs take interleaved
// segments, then add up through LDS (wave 0, in wave order).
__global__ __launch_bounds__(256) void sigma2_combine_kernel(int n, int parts,
                                                             const double* __restrict__ partial,
                                                             double* __restrict__ out) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const double s = i < n ? sigma2_segment_sum(n, parts, partial, (size_t)i, w) : 0.0;
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && i < n) out[i] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

}  // namespace hmg
