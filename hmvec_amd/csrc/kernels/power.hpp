// K6 - one-pair fused 1-halo + 2-halo mass integrals (hmvec/hmvec.py:469-572).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- K6: fused mass integrals (P1-P4)
// Tracer weights are linear forms in at most NT distinct [z][m][k] tensors:
//     form_f(z,m,k) = c[f][0](z,m) + sum_t c[f][1+t](z,m) * T_t(z,m,k)
// f = 0,1: the two factors of the 1-halo integrand (trapz weight and n(z,m) folded into
// factor 0); f = 2,3: the two 2-halo integrands (weight, n and b_h folded in).
// power_prep_kernel builds the coefficient table + the k->0 consistency integrals and
// biases; power_kernel streams every distinct tensor exactly once.
constexpr int PW_NF = 4;
constexpr int PW_MAXT = 4;

struct TracerDev {
    int kind;
    int t_prof, t_cprof;  // slots in the distinct-tensor list, -1 = none
    const double *Nc, *Ns, *NcNs, *NsNsm1, *ngal, *bias_override;
};
struct PowerPrep {
    TracerDev a, b;
    int nt;
    double rho_m0;
};

// Linear form of one tracer's 2-halo weight (also its 1-halo factor in the generic case).
__device__ __forceinline__ void tracer_form(const TracerDev& T, size_t idx, int z, double mass,
                                            double rho_m0, double* c /*[1+PW_MAXT]*/,
                                            double& lowk) {
    for (int i = 0; i <= PW_MAXT; ++i) c[i] = 0.0;
    if (T.kind == HMG_TRACER_MATTER) {
        c[1 + T.t_prof] = mass / rho_m0;
        lowk = mass / rho_m0;
    } else if (T.kind == HMG_TRACER_PRESSURE) {
        c[1 + T.t_prof] = 1.0;
        lowk = 0.0;
    } else {
        const double ng = T.ngal[z], nc = T.Nc[idx], ns = T.Ns[idx];
        if (T.t_cprof >= 0) c[1 + T.t_cprof] += nc / ng; else c[0] += nc / ng;
        c[1 + T.t_prof] += ns / ng;
        lowk = (nc + ns) / ng;
    }
}

// grid nz blocks, 256 threads; coef layout [z][m][PW_NF][1+nt]; side[z][4] = {bA, CA, bB, CB}
__global__ __launch_bounds__(256) void power_prep_kernel(int nm, PowerPrep Q,
                                                         const double* __restrict__ nzm,
                                                         const double* __restrict__ bh,
                                                         const double* __restrict__ ms,
                                                         const double* __restrict__ wm,
                                                         double* __restrict__ coef,
                                                         double* __restrict__ side) {
    __shared__ double lds[16];
    const int z = blockIdx.x;
    const int nc1 = 1 + Q.nt;
    double accCA = 0.0, accCB = 0.0, accBA = 0.0, accBB = 0.0;
    for (int m = threadIdx.x; m < nm; m += blockDim.x) {
        const size_t idx = (size_t)z * nm + m;
        const double mass = ms[m];
        const double wn = wm[m] * nzm[idx];
        const double wnb = wn * bh[idx];
        double fa[1 + PW_MAXT], fb[1 + PW_MAXT], x1[1 + PW_MAXT], x2[1 + PW_MAXT];
        double lowA, lowB;
        tracer_form(Q.a, idx, z, mass, Q.rho_m0, fa, lowA);
        tracer_form(Q.b, idx, z, mass, Q.rho_m0, fb, lowB);
        if (Q.a.kind == HMG_TRACER_HOD && Q.b.kind == HMG_TRACER_HOD) {
            // (2 u_c u_s <NcNs> + <Ns(Ns-1)> u_s^2)/ngal^2 of the FIRST name (hmvec.py:510-511)
            for (int i = 0; i <= PW_MAXT; ++i) x1[i] = x2[i] = 0.0;
            const double ng = Q.a.ngal[z], ng2 = ng * ng;
            x1[1 + Q.a.t_prof] = 1.0;
            const double cc = 2.0 * Q.a.NcNs[idx] / ng2;
            if (Q.a.t_cprof >= 0) x2[1 + Q.a.t_cprof] += cc; else x2[0] += cc;
            x2[1 + Q.a.t_prof] += Q.a.NsNsm1[idx] / ng2;
        } else if (Q.a.kind == HMG_TRACER_PRESSURE && Q.b.kind == HMG_TRACER_PRESSURE) {
            // pk_a**2 — first name only (hmvec.py:512-513)
            for (int i = 0; i <= PW_MAXT; ++i) { x1[i] = fa[i]; x2[i] = fa[i]; }
        } else {
            for (int i = 0; i <= PW_MAXT; ++i) { x1[i] = fa[i]; x2[i] = fb[i]; }
        }
        double* c = coef + idx * (size_t)(PW_NF * nc1);
        for (int i = 0; i < nc1; ++i) {
            c[0 * nc1 + i] = wn * x1[i];
            c[1 * nc1 + i] = x2[i];
            c[2 * nc1 + i] = wnb * fa[i];
            c[3 * nc1 + i] = wnb * fb[i];
        }
        accCA += wnb * lowA;
        accCB += wnb * lowB;
        if (Q.a.kind == HMG_TRACER_HOD) accBA += wnb * (Q.a.Nc[idx] + Q.a.Ns[idx]);
        if (Q.b.kind == HMG_TRACER_HOD) accBB += wnb * (Q.b.Nc[idx] + Q.b.Ns[idx]);
    }
    const double CA = block_sum(accCA, lds), CB = block_sum(accCB, lds);
    const double BA = block_sum(accBA, lds), BB = block_sum(accBB, lds);
    if (threadIdx.x == 0) {
        auto bias = [&](const TracerDev& T, double hodsum) {
            if (T.bias_override) return T.bias_override[z];
            if (T.kind == HMG_TRACER_MATTER) return 1.0;
            if (T.kind == HMG_TRACER_PRESSURE) return 0.0;
            return hodsum / T.ngal[z];
        };
        side[z * 4 + 0] = bias(Q.a, BA);
        side[z * 4 + 1] = CA;
        side[z * 4 + 2] = bias(Q.b, BB);
        side[z * 4 + 3] = CB;
    }
}

struct PowerArgs {
    const double* tens[PW_MAXT];
    const double* coef;
    const double* side;
    const double* ks;
    const double* Pzk;
    double* P1h;
    double* P2h;
    double* I1;       // optional: the two 2-halo integrals I_a(z,k), I_b(z,k) and
    double* I2;
    double* Cout;     // [nz][2] their k -> 0 limits C_a, C_b (get_power_2halo(verbose=True))
    double kstar;
    int nm, nk;
};

template <int V> struct VecT;
template <> struct VecT<1> { using type = double; };
template <> struct VecT<2> { using type = double2; };


template <int V> __device__ __forceinline__ double vget(const typename VecT<V>::type& v, int i);
template <> __device__ __forceinline__ double vget<1>(const double& v, int) { return v; }
template <> __device__ __forceinline__ double vget<2>(const double2& v, int i) { return i ? v.y : v.x; }
template <int V> __device__ __forceinline__ typename VecT<V>::type vsplat(double x);
template <> __device__ __forceinline__ double vsplat<1>(double x) { return x; }
template <> __device__ __forceinline__ double2 vsplat<2>(double x) { return make_double2(x, x); }
// streamed-once tensor data: non-temporal load (does not displace the coefficient rows and hints in the caches)
template <int V> __device__ __forceinline__ typename VecT<V>::type vload_nt(const double* p);
template <> __device__ __forceinline__ double vload_nt<1>(const double* p) { return __builtin_nontemporal_load(p); }
template <> __device__ __forceinline__ double2 vload_nt<2>(const double* p) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
    return make_double2(v.x, v.y);
}


// grid (ceil(nk/(64 V)), nz); block 64*MS threads: lane -> V consecutive k, wave -> an
// interleaved slice of the mass axis.  Each wave streams 512 B*V per tensor per mass bin
// (fully coalesced), partial sums over the MS slices are combined through LDS.
template <int NT, int V>
__global__ __launch_bounds__(1024) void power_kernel(PowerArgs A) {
    extern __shared__ double red[];  // [MS][3][V][64]
    using vec_t = typename VecT<V>::type;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int MS = blockDim.x >> 6;
    const int z = blockIdx.y;
    const int k0 = (blockIdx.x * 64 + lane) * V;
    const bool live = k0 < A.nk;  // nk % V == 0 is guaranteed by the launcher
    constexpr int NC1 = 1 + NT;
    double a1[V], aA[V], aB[V];
#pragma unroll
    for (int v = 0; v < V; ++v) a1[v] = aA[v] = aB[v] = 0.0;
    const size_t zrow = (size_t)z * A.nm;
#pragma unroll 4
    for (int m = wv; m < A.nm; m += MS) {
        const double* __restrict__ c = A.coef + (zrow + m) * (size_t)(PW_NF * NC1);
        vec_t t[NT];
        const size_t off = (zrow + m) * (size_t)A.nk + k0;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (live) t[i] = vload_nt<V>(A.tens[i] + off);
            else t[i] = vec_t{};
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double f0 = c[0 * NC1], f1 = c[1 * NC1], f2 = c[2 * NC1], f3 = c[3 * NC1];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const double tv = vget<V>(t[i], v);
                f0 += c[0 * NC1 + 1 + i] * tv;
                f1 += c[1 * NC1 + 1 + i] * tv;
                f2 += c[2 * NC1 + 1 + i] * tv;
                f3 += c[3 * NC1 + 1 + i] * tv;
            }
            a1[v] += f0 * f1;
            aA[v] += f2;
            aB[v] += f3;
        }
    }
    // combine the MS mass slices
#pragma unroll
    for (int v = 0; v < V; ++v) {
        red[((wv * 3 + 0) * V + v) * 64 + lane] = a1[v];
        red[((wv * 3 + 1) * V + v) * 64 + lane] = aA[v];
        red[((wv * 3 + 2) * V + v) * 64 + lane] = aB[v];
    }
    __syncthreads();
    if (wv == 0 && live) {
        const double bA = A.side[z * 4 + 0], CA = A.side[z * 4 + 1];
        const double bB = A.side[z * 4 + 2], CB = A.side[z * 4 + 3];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double s1 = 0.0, sA = 0.0, sB = 0.0;
            for (int w = 0; w < MS; ++w) {
                s1 += red[((w * 3 + 0) * V + v) * 64 + lane];
                sA += red[((w * 3 + 1) * V + v) * 64 + lane];
                sB += red[((w * 3 + 2) * V + v) * 64 + lane];
            }
            const int k = k0 + v;
            const size_t o = (size_t)z * A.nk + k;
            if (A.P1h) {
                const double q = A.ks[k] / A.kstar;
                A.P1h[o] = s1 * (1.0 - exp(-(q * q)));
            }
            if (A.P2h) A.P2h[o] = A.Pzk[o] * (sA + bA - CA) * (sB + bB - CB);
            if (A.I1) A.I1[o] = sA;
            if (A.I2) A.I2[o] = sB;
            if (A.Cout && k == 0) { A.Cout[z * 2] = CA; A.Cout[z * 2 + 1] = CB; }
        }
    }
}

}  // namespace hmg
