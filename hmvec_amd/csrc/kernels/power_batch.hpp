// K6b - all pairs of a batch in one pass over the tensors (hmvec/hmvec.py:469-572).
// Part of the ONE translation unit hmgrid.hip (included there in this order; not a stand-alone header).
#pragma once

namespace hmg {

// ---------------------------------------------------------------- K6b: all-pairs mass integrals
// When several spectra share profile tensors (Config 3: six spectra over TWO tensors, because
// the galaxy tracer's satellite profile is the NFW tensor), the per-pair kernel re-reads
// each tensor once per pair (sum d = 8 tensor passes).  This kernel takes NTR tracers over NT
// distinct tensors and accumulates, in ONE pass over the tensors, the NTR 2-halo integrals
// I_t and all NTR(NTR+1)/2 1-halo integrals; the 2-halo spectrum of any pair is assembled in
// the epilogue from (I_a, I_b).  Per-tracer forms: W (2-halo weight and cross 1-halo factor)
// and (A1, A2), the two factors of the tracer's 1-halo AUTO integrand (= W, W except for an
// HOD, whose auto term is (2 u_c u_s <NcNs> + <Ns(Ns-1)> u_s^2)/ngal^2).
constexpr int PB_MAXTR = 4;
constexpr int PB_MAXPAIR = PB_MAXTR * (PB_MAXTR + 1) / 2;

// Structure of a batch.  Most coefficients of the generic linear forms are structural zeros or ones:
//   matter / pressure on tensor s   ("LIN s"):  W = A1 = A2 = c t_s                          1 number per (z,m)
//   HOD, satellites on s, u_c == 1  ("HOD s"):  W = c0 + c1 t_s, A1 = t_s, A2 = a0 + a1 t_s  4 numbers
// For the batches the facade issues most (PB_SPEC_LIST) the kernel is compiled for that structure: a row of
// 2 + sum numbers instead of 2 + 3 NTR (1+NT), padded to whole 64-byte lines - Config 3: 8 doubles, one
// s_load_dwordx16, against 29 - and only the non-zero terms are evaluated, with the operations the generic
// forms apply to them (adding an exact zero or multiplying by an exact one changes no bit), so both paths
// give the same sums.  What the mass loop cannot afford is scalar-memory traffic per bin.
// A code packs 4 bits per tracer, tracer 0 lowest: bits 0-1 kind (0 none, 1 LIN, 2 HOD), bits 2-3 tensor slot.
constexpr unsigned PB_LIN(int s) { return 1u | ((unsigned)s << 2); }
constexpr unsigned PB_HOD(int s) { return 2u | ((unsigned)s << 2); }
constexpr unsigned pb_code(unsigned t0, unsigned t1 = 0, unsigned t2 = 0, unsigned t3 = 0) {
    return t0 | (t1 << 4) | (t2 << 8) | (t3 << 12);
}
constexpr int pb_kind(unsigned code, int r) { return (int)((code >> (4 * r)) & 3u); }
constexpr int pb_slot(unsigned code, int r) { return (int)((code >> (4 * r + 2)) & 3u); }
constexpr int pb_ncoef(unsigned code, int ntr) {     // numbers per compact row before padding
    int n = 2;
    for (int r = 0; r < ntr; ++r) n += pb_kind(code, r) == 2 ? 4 : 1;
    return n;
}
constexpr int pb_stride(unsigned code, int ntr, int nc1) {   // doubles per (z,m) coefficient row
    return code ? ((pb_ncoef(code, ntr) + 7) & ~7) : 2 + ntr * 3 * nc1;
}

// (distinct tensors, tracers, structure) the kernel is compiled for: what get_power*/spectra_block produce for
// matter and pressure profiles and an HOD whose satellites follow the first matter profile (the reference's
// README usage: 'nfw', a Battaglia gas profile, a pressure profile, an HOD)
#define PB_SPEC_LIST                                                  \
    PB_SPEC(1, 1, PB_LIN(0))                                          \
    PB_SPEC(1, 2, PB_LIN(0), PB_HOD(0))                               \
    PB_SPEC(2, 2, PB_LIN(0), PB_LIN(1))                               \
    PB_SPEC(2, 3, PB_LIN(0), PB_LIN(1), PB_HOD(0))                    \
    PB_SPEC(3, 3, PB_LIN(0), PB_LIN(1), PB_LIN(2))                    \
    PB_SPEC(3, 4, PB_LIN(0), PB_LIN(1), PB_LIN(2), PB_HOD(0))

struct BatchPrep {
    TracerDev tr[PB_MAXTR];
    int ntr, nt;
    double rho_m0;
    unsigned code;       // 0: generic rows; else the compact rows of that structure
};

// coef layout per (z,m): [wn, wnb, {W[1+nt], A1[1+nt], A2[1+nt]} x ntr].
// grid (nz, nblk) with 64-thread blocks, one (z,m) per thread; the k->0 consistency sums C_t
// and HOD bias numerators B_t are written as per-block partials sidep[z][blk][t][2] = {B, C}
// and summed in block order by the main kernel's epilogue (deterministic).
struct PrepArgs {
    int nm, nblk;
    BatchPrep Q;
    const double *nzm, *bh, *ms, *wm;
    double *coef, *sidep;
};
// one wavefront = the 64 masses of tile blk of redshift z
__device__ __forceinline__ void batch_prep_tile(const PrepArgs& PA, int z, int blk) {
    const int nm = PA.nm, nblk = PA.nblk;
    const BatchPrep& Q = PA.Q;
    const double* __restrict__ nzm = PA.nzm;
    const double* __restrict__ bh = PA.bh;
    const double* __restrict__ ms = PA.ms;
    const double* __restrict__ wm = PA.wm;
    double* __restrict__ coef = PA.coef;
    double* __restrict__ sidep = PA.sidep;
    const int m = blk * 64 + (threadIdx.x & 63);
    const int nc1 = 1 + Q.nt;
    const int stride = pb_stride(Q.code, Q.ntr, nc1);
    double accC[PB_MAXTR], accB[PB_MAXTR];
    for (int t = 0; t < PB_MAXTR; ++t) accC[t] = accB[t] = 0.0;
    if (m < nm) {
        const size_t idx = (size_t)z * nm + m;
        const double mass = ms[m];
        const double wn = wm[m] * nzm[idx];
        const double wnb = wn * bh[idx];
        double* c = coef + idx * (size_t)stride;
        c[0] = wn;
        c[1] = wnb;
        int pos = 2;
        for (int t = 0; t < Q.ntr; ++t) {
            const TracerDev& T = Q.tr[t];
            double w[1 + PW_MAXT], a1[1 + PW_MAXT], a2[1 + PW_MAXT], low;
            tracer_form(T, idx, z, mass, Q.rho_m0, w, low);
            for (int i = 0; i <= PW_MAXT; ++i) { a1[i] = w[i]; a2[i] = w[i]; }
            if (T.kind == HMG_TRACER_HOD) {
                for (int i = 0; i <= PW_MAXT; ++i) a1[i] = a2[i] = 0.0;
                const double ng = T.ngal[z], ng2 = ng * ng;
                a1[1 + T.t_prof] = 1.0;
                const double cc = 2.0 * T.NcNs[idx] / ng2;
                if (T.t_cprof >= 0) a2[1 + T.t_cprof] += cc; else a2[0] += cc;
                a2[1 + T.t_prof] += T.NsNsm1[idx] / ng2;
                accB[t] = wnb * (T.Nc[idx] + T.Ns[idx]);
            }
            if (Q.code == 0) {
                double* ct = c + 2 + t * 3 * nc1;
                for (int i = 0; i < nc1; ++i) {
                    ct[i] = w[i];
                    ct[nc1 + i] = a1[i];
                    ct[2 * nc1 + i] = a2[i];
                }
            } else {                  // compact row: only the numbers that are not structural zeros / ones
                const int sl = 1 + pb_slot(Q.code, t);
                if (pb_kind(Q.code, t) == 1) {
                    c[pos++] = w[sl];
                } else {
                    c[pos++] = w[0]; c[pos++] = w[sl]; c[pos++] = a2[0]; c[pos++] = a2[sl];
                }
            }
            accC[t] = wnb * low;
        }
        if (Q.code) for (; pos < stride; ++pos) c[pos] = 0.0;
    }
    for (int t = 0; t < Q.ntr; ++t) {
        const double C = wave_sum(accC[t]);
        const double B = wave_sum(accB[t]);
        if ((threadIdx.x & 63) == 0) {
            double* sp = sidep + ((size_t)(z * nblk + blk) * Q.ntr + t) * 2;
            sp[0] = B;
            sp[1] = C;
        }
    }
}
// The same rows for a batch with a structure code (every batch of PB_SPEC_LIST), written without the
// generic forms' dynamically indexed coefficient arrays: a handful of registers, so that it can run as a
// link of the per-z chain inside the profile group under that kernel's 64-register budget without spilling
// (a spill anywhere gives the whole launch a scratch allocation, which cost the fused profile rows 7 %).
// Same numbers as batch_prep_tile: the generic forms add these terms to exact zeros.
__device__ __forceinline__ void batch_prep_tile_compact(const PrepArgs& PA, int z, int blk) {
    const BatchPrep& Q = PA.Q;
    const int nm = PA.nm, lane = threadIdx.x & 63, m = blk * 64 + lane;
    int ncoef = 2;
#pragma unroll
    for (int t = 0; t < PB_MAXTR; ++t)
        if (t < Q.ntr) ncoef += Q.tr[t].kind == HMG_TRACER_HOD ? 4 : 1;
    const int stride = (ncoef + 7) & ~7;
    double accC[PB_MAXTR], accB[PB_MAXTR];
#pragma unroll
    for (int t = 0; t < PB_MAXTR; ++t) accC[t] = accB[t] = 0.0;
    if (m < nm) {
        const size_t idx = (size_t)z * nm + m;
        const double mass = PA.ms[m];
        const double wn = PA.wm[m] * PA.nzm[idx];
        const double wnb = wn * PA.bh[idx];
        double* __restrict__ c = PA.coef + idx * (size_t)stride;
        c[0] = wn;
        c[1] = wnb;
        int pos = 2;
#pragma unroll
        for (int t = 0; t < PB_MAXTR; ++t) {
            if (t >= Q.ntr) continue;
            const TracerDev& T = Q.tr[t];
            double low = 0.0;
            if (T.kind == HMG_TRACER_HOD) {
                const double ng = T.ngal[z], ng2 = ng * ng, nc = T.Nc[idx], ns = T.Ns[idx];
                c[pos] = nc / ng;
                c[pos + 1] = ns / ng;
                c[pos + 2] = 2.0 * T.NcNs[idx] / ng2;
                c[pos + 3] = T.NsNsm1[idx] / ng2;
                pos += 4;
                accB[t] = wnb * (nc + ns);
                low = (nc + ns) / ng;
            } else {
                low = T.kind == HMG_TRACER_MATTER ? mass / Q.rho_m0 : 0.0;
                c[pos++] = T.kind == HMG_TRACER_MATTER ? low : 1.0;
            }
            accC[t] = wnb * low;
        }
        for (; pos < stride; ++pos) c[pos] = 0.0;
    }
#pragma unroll
    for (int t = 0; t < PB_MAXTR; ++t) {
        if (t >= Q.ntr) continue;
        const double C = wave_sum(accC[t]);
        const double B = wave_sum(accB[t]);
        if (lane == 0) {
            double* sp = PA.sidep + ((size_t)(z * PA.nblk + blk) * Q.ntr + t) * 2;
            sp[0] = B;
            sp[1] = C;
        }
    }
}
__global__ __launch_bounds__(64) void power_batch_prep_kernel(PrepArgs PA) {
    if (PA.Q.code) batch_prep_tile_compact(PA, blockIdx.x, blockIdx.y);
    else batch_prep_tile(PA, blockIdx.x, blockIdx.y);
}

struct BatchArgs {
    const double* tens[PW_MAXT];
    const int* nconst[PW_MAXT];      // constant-prefix hint of tensor i ([nz][nm]) or nullptr
    const double* cconst[PW_MAXT];
    const double* coef;
    const double* sidep;             // [nz][nblk][NTR][2] partial {B, C}
    const double* ngal[PB_MAXTR];    // HOD tracers: ngal[z] (bias = B/ngal); else nullptr
    double bias_const[PB_MAXTR];     // matter 1, pressure 0
    int nblk;
    const double* ks;
    const double* Pzk;
    double* P1h[PB_MAXPAIR];  // canonical pair index of (a<=b): a*NTR - a(a-1)/2 + (b-a)
    double* P2h[PB_MAXPAIR];
    double kstar;
    int nm, nk;
};

// Summation order over the mass axis (fixed by nm alone, so that a z-slab run and the full grid agree
// bit for bit whatever launch shape each picks): PB_NV = 16 virtual slices, slice v = the bins
// m = v, v+16, v+32, ... summed in that order from zero; then the pair sums t_w = s_w + s_{w+8};
// then t_0 + t_1 + ... + t_7 in that order.  Two launch shapes realise it:
//   W16 = false: 8 wavefronts (512 threads), wavefront w walks slice w, parks the sums in its private
//                part of LDS (no barrier), walks slice w+8 and adds the parked sums at the end;
//   W16 = true : 16 wavefronts (1024 threads), one slice each - twice the loads in flight per CU,
//                which is what a thin z-slab (one workgroup per CU) needs.
constexpr int PB_NV = 16;

// the value lane l of the wavefront holds (l uniform): two v_readlane_b32
__device__ __forceinline__ double lane_value(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int NT, int NTR, int V, bool W16, unsigned CODE = 0>
__global__ __launch_bounds__(W16 ? 1024 : 512) void power_batch_kernel(BatchArgs A) {
    extern __shared__ double red[];  // [8][NACC*V][64]: parked sums / pair exchange, then the cross-wave reduction
    using vec_t = typename VecT<V>::type;
    constexpr int NC1 = 1 + NT;
    constexpr int STRIDE = pb_stride(CODE, NTR, NC1);
    constexpr int NPAIR = NTR * (NTR + 1) / 2;
    constexpr int NACC = NTR + NPAIR;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z = blockIdx.y;
    const int k0 = (blockIdx.x * 64 + lane) * V;
    const bool live = k0 < A.nk;
    double acc[NACC][V];      // [0, NTR): 2-halo integrals I_t; [NTR, NACC): 1-halo integrals of the pairs
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[a][v] = 0.0;
    const size_t zrow = (size_t)z * A.nm;
    const int kend = min(A.nk, (int)(blockIdx.x + 1) * 64 * V);   // one past the last k of this tile
    const size_t kofs = live ? (size_t)k0 : 0;   // dead lanes re-read column 0 (their sums are never stored)
    // the bins of this wavefront, in order: position i -> mass bin (>= nm: no such bin)
    const int L = (A.nm + PB_NV - 1) / PB_NV;            // positions per slice
    const int NB = W16 ? L : 2 * L;
    auto bin = [&](int i) {
        if (W16) return wv + PB_NV * i;
        return i < L ? wv + PB_NV * i : wv + 8 + PB_NV * (i - L);
    };
    // constant-prefix hint of mass bin m: how many leading k of the row equal `val`
    struct Hint { int n[NT]; double val[NT]; };
    auto load_hint = [&](Hint& h, int m) {
        const size_t r = zrow + min(m, A.nm - 1);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            h.n[i] = A.nconst[i] ? A.nconst[i][r] : -1;
            h.val[i] = A.nconst[i] ? A.cconst[i][r] : 0.0;
        }
    };
    // rows whose whole k tile lies in a tensor's constant prefix are not read at all
    auto fetch = [&](vec_t (&dst)[NT], int m, const Hint& h) {
        const size_t off = (zrow + min(m, A.nm - 1)) * (size_t)A.nk + kofs;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (h.n[i] >= kend) dst[i] = vsplat<V>(h.val[i]);
            else dst[i] = vload_nt<V>(A.tens[i] + off);
        }
    };
    auto park = [&]() {      // end of the first slice (8-wavefront shape): sums to LDS, start again from zero
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                red[((wv * NACC + a) * V + v) * 64 + lane] = acc[a][v];
                acc[a][v] = 0.0;
            }
    };
    auto accumulate = [&](const vec_t (&t)[NT], int i) {
        if (!W16 && i == L) park();
        const int m = bin(i);
        if (m >= A.nm) return;
        const double* __restrict__ c = A.coef + (zrow + m) * (size_t)STRIDE;
        const double wn = c[0], wnb = c[1];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double W[NTR], A1[NTR], A2[NTR];
            if constexpr (CODE != 0) {
                // compiled for this batch's structure: only the non-zero terms of the forms
                int pos = 2;
#pragma unroll
                for (int r = 0; r < NTR; ++r) {
                    const double ts = vget<V>(t[pb_slot(CODE, r)], v);
                    if (pb_kind(CODE, r) == 1) {          // (constant after unrolling)
                        W[r] = c[pos] * ts;
                        A1[r] = W[r]; A2[r] = W[r];
                        pos += 1;
                    } else {
                        W[r] = fma(c[pos + 1], ts, c[pos]);
                        A1[r] = ts;
                        A2[r] = fma(c[pos + 3], ts, c[pos + 2]);
                        pos += 4;
                    }
                    acc[r][v] += wnb * W[r];
                }
            } else
#pragma unroll
            for (int r = 0; r < NTR; ++r) {
                const double* cr = c + 2 + r * 3 * NC1;
                double w = cr[0], a1 = cr[NC1], a2 = cr[2 * NC1];
#pragma unroll
                for (int i2 = 0; i2 < NT; ++i2) {
                    const double tv = vget<V>(t[i2], v);
                    w += cr[1 + i2] * tv;
                    a1 += cr[NC1 + 1 + i2] * tv;
                    a2 += cr[2 * NC1 + 1 + i2] * tv;
                }
                W[r] = w; A1[r] = a1; A2[r] = a2;
                acc[r][v] += wnb * w;
            }
            int p = NTR;
#pragma unroll
            for (int a = 0; a < NTR; ++a) {
                acc[p][v] += wn * (A1[a] * A2[a]);
                ++p;
#pragma unroll
                for (int b = a + 1; b < NTR; ++b) {
                    acc[p][v] += wn * (W[a] * W[b]);
                    ++p;
                }
            }
        }
    };
    // Two-stage software pipeline over this wavefront's bins: the loads of the next bin are in flight
    // while the current one is consumed, and the hints run one bin further ahead so that a fetch never
    // waits for its own decision.  The scheduling barriers keep hipcc from sinking the early loads back
    // down to their first use.
    int i = 0;
    // (Round 3, thin z-slabs: a four-stage version of this pipeline - three bins of tensor loads in flight - was
    // measured on the 16-wavefront shapes: 25.9 -> 26.0 us at nz = 4, 39.9 -> 40.7 at nz = 8.  What a wavefront
    // waits for there is the scalar load of the next bin's coefficient row, which cannot run ahead: two rows do
    // not fit the scalar register file.  Staging each wavefront's rows in LDS a chunk ahead - vector loads in
    // flight during the previous chunk, coefficients read by broadcast ds_read_b64 - was also built: bit-identical
    // and slower, 26.4 -> 39.8 us at nz = 4 and 41.4 -> 45.0 at nz = 8, since 29 LDS reads per bin and wavefront
    // occupy the LDS pipe for longer than the scalar round trip they replace.  Fetching the (8-double, structure-
    // compiled) coefficient row one bin ahead with the tensors: 26.8 -> 26.9 us at nz = 4, 134.6 -> 138.2 at nz = 32.
    // The thin launch moves its 97 MB at 3.7 TB/s with every CU holding ~32 KB of loads in flight, the same
    // per-CU amount all shapes of this kernel reach: it is the memory system's latency.)
    vec_t ta[NT], tb[NT];
    Hint ha, hb;
    load_hint(ha, bin(0));
    load_hint(hb, bin(1));
    fetch(ta, bin(0), ha);
#pragma unroll 1
    for (; i + 1 < NB; i += 2) {
        fetch(tb, bin(i + 1), hb);
        load_hint(ha, bin(i + 2));
        __builtin_amdgcn_sched_barrier(0);
        accumulate(ta, i);
        __builtin_amdgcn_sched_barrier(0);
        fetch(ta, bin(i + 2), ha);
        load_hint(hb, bin(i + 3));
        __builtin_amdgcn_sched_barrier(0);
        accumulate(tb, i + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (i < NB) accumulate(ta, i);
    // pair sums t_w = s_w + s_{w+8}
    if (W16) {
        if (wv >= 8) {
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
                for (int v = 0; v < V; ++v) red[(((wv - 8) * NACC + a) * V + v) * 64 + lane] = acc[a][v];
        }
        __syncthreads();
    }
    if (wv < 8) {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double other = red[((wv * NACC + a) * V + v) * 64 + lane];
                acc[a][v] = W16 ? acc[a][v] + other : other + acc[a][v];     // s_w + s_{w+8}
            }
    }
    // ordered sum over the eight pair sums through LDS, one accumulator at a time (the parked values have
    // been consumed: the same memory serves as [8][V][64] exchange buffer)
    auto reduce = [&](double (&x)[V]) {
        __syncthreads();
        if (wv < 8) {
#pragma unroll
            for (int v = 0; v < V; ++v) red[(wv * V + v) * 64 + lane] = x[v];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                double sum = 0.0;
                for (int w = 0; w < 8; ++w) sum += red[(w * V + v) * 64 + lane];
                x[v] = sum;
            }
        }
    };
#pragma unroll
    for (int a = 0; a < NACC; ++a) reduce(acc[a]);
    if (wv == 0) {
        // what the outputs need besides the integrals is requested FIRST, so that it is in flight during the sums below:
        // the wavenumbers and P_lin of this tile and the 2 NPAIR output pointers (one block of the argument segment
        // instead of one scalar load, wait and branch per output)
        double kv[V], pl[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            kv[v] = live ? A.ks[k0 + v] : 1.0;
            pl[v] = (live && A.Pzk) ? A.Pzk[(size_t)z * A.nk + k0 + v] : 0.0;
        }
        double* o1[NPAIR];
        double* o2[NPAIR];
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) { o1[p] = A.P1h[p]; o2[p] = A.P2h[p]; }
        // Consistency sums B_t, C_t of the batch's tracers over the 64-mass tiles, in tile order.  One coalesced load
        // brings a group of whole tiles (64 values) into the lanes and the ordered sums read them out with
        // v_readlane: as a loop of scalar loads this was one dependent scalar-memory round trip per tile and tracer (24
        // for Config 3, ~2.7 us) at the tail of every workgroup - on a thin z-slab, of the launch.  Same order of
        // additions, same bits.  (All 64 lanes take part: a lane without a live wavenumber still holds its value.)
        double bmc[NTR];  // b_t - C_t
        {
            double Bs[NTR], Cs[NTR];
#pragma unroll
            for (int t = 0; t < NTR; ++t) Bs[t] = Cs[t] = 0.0;
            constexpr int CH = 64 / (2 * NTR);                      // whole tiles per trip
            const double* __restrict__ spz = A.sidep + (size_t)z * A.nblk * (NTR * 2);
            for (int blk0 = 0; blk0 < A.nblk; blk0 += CH) {
                const int nb = min(CH, A.nblk - blk0);
                const double val = lane < nb * NTR * 2 ? spz[(size_t)blk0 * (NTR * 2) + lane] : 0.0;
                for (int b = 0; b < nb; ++b) {
#pragma unroll
                    for (int t = 0; t < NTR; ++t) {
                        Bs[t] += lane_value(val, (b * NTR + t) * 2);
                        Cs[t] += lane_value(val, (b * NTR + t) * 2 + 1);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NTR; ++t) {
                const double bias = A.ngal[t] ? Bs[t] / A.ngal[t][z] : A.bias_const[t];
                bmc[t] = bias - Cs[t];
            }
        }
        if (live) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int k = k0 + v;
                const size_t o = (size_t)z * A.nk + k;
                const double q = kv[v] / A.kstar;
                const double damp = 1.0 - exp(-(q * q));
                const double plin = pl[v];
                int p = 0;
#pragma unroll
                for (int a = 0; a < NTR; ++a) {
#pragma unroll
                    for (int b = a; b < NTR; ++b) {
                        if (o1[p]) o1[p][o] = acc[NTR + p][v] * damp;
                        // (the two brackets are multiplied first: commutative, so the result does not depend on
                        // which of the two tracers got the lower index in this batch)
                        if (o2[p]) o2[p][o] = plin * ((acc[a][v] + bmc[a]) * (acc[b][v] + bmc[b]));
                        ++p;
                    }
                }
            }
        }
    }
}

}  // namespace hmg
