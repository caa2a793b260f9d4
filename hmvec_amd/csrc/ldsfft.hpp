// Workgroup-local mixed-radix FFT used by the fused radial-profile kernel.
//
// The reference transforms every (z,m) row with np.fft.rfft (hmvec/fft.py:49) and keeps only
// the imaginary part.  Here one workgroup owns one row: the N real samples are packed as
// M = N/2 complex numbers in LDS, transformed in place by an autosort (Stockham) FFT with
// radix-5/4/3/2 passes, and unpacked to Im F_j on the fly, so the row never leaves the CU.
//
// The per-thread pieces are plain functions of (thread index, buffer pointer) so the same
// code is compiled for the GPU (buffer = LDS, a barrier between the load and store halves
// of each pass) and for the host, where tests/test_ldsfft_cpu.py runs it thread by thread
// against numpy.
#pragma once

#include <vector>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HMG_HD __host__ __device__ __forceinline__
#else
#define HMG_HD inline
#endif

namespace hmg {

struct alignas(16) cplx {
    double x, y;
};
HMG_HD cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
HMG_HD cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
HMG_HD cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
// multiply by -i  (forward-transform rotation)
HMG_HD cplx cmul_mi(cplx a) { return {a.y, -a.x}; }

constexpr int FFT_MAX_PASSES = 16;
struct FftPlanDev {
    int M;       // complex length
    int npass;
    int radix[FFT_MAX_PASSES];
    // per pass: Ns = product of the earlier radices, the twiddle-table step M/(Ns*R), and the
    // multiplier that turns j / Ns into one 32x32 high multiply (exact for j < 2^16; 0 when Ns == 1)
    int ns[FFT_MAX_PASSES];
    int twstep[FFT_MAX_PASSES];
    unsigned magic[FFT_MAX_PASSES];
    int twoff[FFT_MAX_PASSES];    // start of the pass's slice in the per-pass twiddle table (pass_tw_table): sum of the earlier Ns
};

// j / d for 0 <= j < 2^16, 1 < d < 2^16 with magic = floor(2^32 / d) + 1 (magic = 0 encodes d == 1)
HMG_HD unsigned fast_div(unsigned j, unsigned magic) {
#if defined(__HIP_DEVICE_COMPILE__)
    return magic ? __umulhi(j, magic) : j;
#else
    return magic ? (unsigned)(((unsigned long long)j * magic) >> 32) : j;
#endif
}

// The same for butterflies of a short transform (j < 1024, d <= 1024) with magic = ceil(2^20 / d): the product
// stays below 2^30 and the operands below 2^24, so that the GPU's full-rate 24-bit multiplier does it
// (v_mul_hi_u32 / v_mul_lo_u32 run at a quarter of that rate).  Exact: j * (magic * d - 2^20) < j * d <= 2^20.
HMG_HD unsigned mul_small(unsigned a, unsigned b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
constexpr unsigned small_magic(unsigned d) { return d == 1 ? 0u : ((1u << 20) + d - 1) / d; }
template <bool SMALL>
HMG_HD unsigned div_ns(unsigned j, unsigned magic) {
    if constexpr (SMALL) return magic ? mul_small(j, magic) >> 20 : j;
    else return fast_div(j, magic);
}
template <bool SMALL>
HMG_HD int mul_idx(int a, int b) {
    if constexpr (SMALL) return (int)mul_small((unsigned)a, (unsigned)b);
    else return a * b;
}

// Factor M into radices 5,4,3,2 (as few passes as possible: 4 before 2), then order the passes by
// ascending radix.  The first pass is the one the fused profile kernel can skip when the input is
// zero beyond sample M/R0 (truncated profiles): the smallest radix gives the loosest condition and
// is the pass with the most butterflies.  Returns false if a larger prime remains.
// constexpr: the kernels that know their length at compile time (SPECM, the sub-transforms of the pruned
// long-grid route) evaluate the plan in the compiler, so that strides, twiddle steps and multipliers are immediates.
constexpr bool fft_plan_fill(int M, FftPlanDev& p) {
    p.M = M;
    p.npass = 0;
    int rem = M;
    const int cand[4] = {5, 4, 3, 2};
    for (int ci = 0; ci < 4; ++ci) {
        const int r = cand[ci];
        while (rem % r == 0 && rem > 1) {
            if (p.npass >= FFT_MAX_PASSES) return false;
            p.radix[p.npass++] = r;
            rem /= r;
        }
    }
    for (int i = 1; i < p.npass; ++i)          // insertion sort, ascending
        for (int j = i; j > 0 && p.radix[j - 1] > p.radix[j]; --j) {
            const int tmp = p.radix[j];
            p.radix[j] = p.radix[j - 1];
            p.radix[j - 1] = tmp;
        }
    // A power-of-two pass right behind a power-of-two first pass stores 16-byte elements at q*(Ns R) + k + t*Ns with
    // Ns R = 8 or 16: the eight lanes of a ds_write_b128 group then fall on two or four 16-byte slots modulo 128 B (a
    // 4-way bank conflict for M = 1000: 2,4,...).  With a radix-5 pass in that place the blocks are 10 or 20 elements
    // apart and the eight lanes cover all eight slots.  The order of the passes behind the first is free.  (No build
    // switch here: the host's twiddle tables and the compile-time SubPass plans of BOTH kernel units come from this one
    // function and must agree.)
    if (p.npass >= 3 && (p.radix[0] == 2 || p.radix[0] == 4) && (p.radix[1] == 2 || p.radix[1] == 4)) {
        for (int i = 2; i < p.npass; ++i)
            if (p.radix[i] == 5) {
                for (int j = i; j > 1; --j) p.radix[j] = p.radix[j - 1];
                p.radix[1] = 5;
                break;
            }
    }
    int Ns = 1;
    for (int i = 0; i < p.npass; ++i) {
        p.ns[i] = Ns;
        p.twstep[i] = M / (Ns * p.radix[i]);
        p.magic[i] = Ns == 1 ? 0u : (unsigned)(4294967296ull / (unsigned)Ns) + 1u;
        p.twoff[i] = i ? p.twoff[i - 1] + p.ns[i - 1] : 0;
        Ns *= p.radix[i];
    }
    return rem == 1 && M >= 2 && M < 65536;
}
inline bool fft_make_plan(int M, FftPlanDev* p) { return fft_plan_fill(M, *p); }
constexpr FftPlanDev fft_plan_c(int M) {
    FftPlanDev p{};
    fft_plan_fill(M, p);
    return p;
}

// Per-pass twiddle tables.  Pass ps of a length-M plan multiplies by W_M^(k twstep), k = j mod Ns < Ns.  Read from the
// one table exp(-2 pi i t / M) that is a gather at a stride of twstep 16-byte elements: for the passes in the middle of a
// plan (M = 1000: Ns = 40, twstep = 5) every lane of a wavefront touches a cache line of its own and the load takes the
// CU's texture-address unit 64 cycles instead of 16 (measured on MI355X in round 5: the uncoalesced table reads of the
// long-grid kernel, not its arithmetic, were its largest single cost).  Laid out per pass - [sum of the earlier Ns + k] -
// consecutive butterflies read consecutive elements; the whole set has sum(Ns) < M/2 entries.
constexpr int pass_tw_offset(const FftPlanDev& p, int ps) {
    int o = 0;
    for (int i = 0; i < ps; ++i) o += p.ns[i];
    return o;
}
constexpr int pass_tw_total(const FftPlanDev& p) { return pass_tw_offset(p, p.npass); }

// In-place forward DFTs of size R (sign -).
template <int R>
HMG_HD void dft_small(cplx* v);

template <>
HMG_HD void dft_small<2>(cplx* v) {
    const cplx a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <>
HMG_HD void dft_small<3>(cplx* v) {
    const double c = -0.5, s = 0.86602540378443864676;  // cos, sin of 2pi/3
    const cplx t1 = cadd(v[1], v[2]), t2 = csub(v[1], v[2]);
    const cplx m = {v[0].x + c * t1.x, v[0].y + c * t1.y};
    const cplx n = {s * t2.x, s * t2.y};
    v[0] = cadd(v[0], t1);
    v[1] = cadd(m, cmul_mi(n));
    v[2] = csub(m, cmul_mi(n));
}
template <>
HMG_HD void dft_small<4>(cplx* v) {
    const cplx t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    const cplx t2 = cadd(v[1], v[3]), t3 = cmul_mi(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
}
template <>
HMG_HD void dft_small<5>(cplx* v) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;  // cos 2pi/5, 4pi/5
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;   // sin 2pi/5, 4pi/5
    const cplx t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    const cplx t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const cplx m1 = {v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y};
    const cplx m2 = {v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y};
    const cplx n1 = cmul_mi(cplx{s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y});
    const cplx n2 = cmul_mi(cplx{s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y});
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(m1, n1);
    v[4] = csub(m1, n1);
    v[2] = cadd(m2, n2);
    v[3] = csub(m2, n2);
}

// DFT_5 of (v0, v1, v2, 0, 0): dft_small<5> with the terms that are exactly zero left out - the same products and
// sums in the same order, so the same values (a butterfly of the pass behind a pruned first pass sees this input
// when the row is zero beyond sample 3M/(R0 R1): see the fused profile kernel).
HMG_HD void dft5_lead3(cplx* v) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    const cplx t1 = v[1], t2 = v[2];
    const cplx m1 = {v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y};
    const cplx m2 = {v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y};
    const cplx n1 = cmul_mi(cplx{s1 * t1.x + s2 * t2.x, s1 * t1.y + s2 * t2.y});
    const cplx n2 = cmul_mi(cplx{s2 * t1.x - s1 * t2.x, s2 * t1.y - s1 * t2.y});
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(m1, n1);
    v[4] = csub(m1, n1);
    v[2] = cadd(m2, n2);
    v[3] = csub(m2, n2);
}

inline std::vector<cplx> pass_tw_table(const FftPlanDev& p) {
    std::vector<cplx> t((size_t)pass_tw_total(p));
    const long double twopi = 6.283185307179586476925286766559L;
    for (int ps = 0; ps < p.npass; ++ps)
        for (int k = 0; k < p.ns[ps]; ++k) {
            const long double a = twopi * (long double)((long long)k * p.twstep[ps]) / (long double)p.M;
            t[(size_t)pass_tw_offset(p, ps) + k] = cplx{(double)cosl(a), (double)-sinl(a)};
        }
    return t;
}

// One Stockham pass of radix R on butterfly j (0 <= j < M/R), sub-transform size Ns so far:
//   load:  v[t] = buf[j + t*M/R] * W_M^(t * k * M/(Ns*R)),  k = j mod Ns  (no integer division: fast_div)
//   store: buf[(j div Ns)*Ns*R + k + t*Ns] = DFT_R(v)[t]
// Every load of a pass must precede every store of that pass (barrier on the GPU).
// (Ns, twstep, magic) are the pass's entries of FftPlanDev.
// SMALL: j < 1024, Ns <= 1024 and magic = small_magic(Ns) (see div_ns).
// NIN < R: only the first NIN inputs of every butterfly are non-zero (R == 5, NIN == 3 only).
// SRC_SHIFT = s: input slot i is read from buf[i >> s] - the pass behind a pruned radix-2^s first pass, whose
// output slot i is sample i >> s, reads the samples themselves (nothing has to be replicated first).
template <int R, bool SMALL = false, int NIN = R, int SRC_SHIFT = 0>
HMG_HD void pass_load(const cplx* buf, const cplx* twM, int M, int Ns, int twstep, unsigned magic, int j, cplx* v) {
    const int k = j - mul_idx<SMALL>((int)div_ns<SMALL>((unsigned)j, magic), Ns);      // j mod Ns
    const int stride = M / R;
    v[0] = buf[j >> SRC_SHIFT];
    // one table read (w = W^(k M/(Ns R)): element k of the pass's slice when the caller hands in a per-pass table and
    // twstep = 1); the higher powers by complex multiplication (<= 3 products, a few ulp) instead of R-1 trips to the
    // table (measured on MI355X in round 3, all powers read: the fused profile kernel goes from 0.204 to 0.245 ms - even
    // a coalesced 16-byte-per-lane load occupies the CU's one vector-memory path for 16 cycles, the 12 multiply-adds it
    // saves cost 48 cycles on one of four SIMDs).
    // k == 0 is not special-cased: its twiddle is twM[0] = 1 exactly, and a branch would make
    // every wavefront that holds such a lane walk both paths.
    const cplx w1 = twM[mul_idx<SMALL>(k, twstep)];
    cplx w = w1;
#pragma unroll
    for (int t = 1; t < NIN; ++t) {
        v[t] = cmul(buf[(j + t * stride) >> SRC_SHIFT], w);
        if (t + 1 < NIN) w = cmul(w, w1);
    }
}
// pass_load with the butterfly's twiddle w1 = W^(k M/(Ns R)) supplied by the caller (a kernel that runs the same pass
// on many transforms keeps it in a register instead of fetching it from the table every time)
template <int R, bool SMALL = false>
HMG_HD void pass_load_w(const cplx* buf, int M, int j, cplx w1, cplx* v) {
    const int stride = M / R;
    v[0] = buf[j];
    cplx w = w1;
#pragma unroll
    for (int t = 1; t < R; ++t) {
        v[t] = cmul(buf[j + t * stride], w);
        if (t + 1 < R) w = cmul(w, w1);
    }
}
template <int R, bool SMALL = false>
HMG_HD int pass_twiddle_index(int Ns, int twstep, unsigned magic, int j) {
    const int k = j - mul_idx<SMALL>((int)div_ns<SMALL>((unsigned)j, magic), Ns);      // j mod Ns
    return mul_idx<SMALL>(k, twstep);
}
template <int R, bool SMALL = false, int NIN = R>
HMG_HD void pass_store(cplx* buf, int Ns, unsigned magic, int j, cplx* v) {
    static_assert(NIN == R || (R == 5 && NIN == 3), "only the 3-of-5 butterfly exists");
    if constexpr (NIN == R) dft_small<R>(v);
    else dft5_lead3(v);
    const int q = (int)div_ns<SMALL>((unsigned)j, magic);          // j div Ns
    const int k = j - mul_idx<SMALL>(q, Ns);
    const int j0 = mul_idx<SMALL>(q, Ns * R) + k;
#pragma unroll
    for (int t = 0; t < R; ++t) buf[j0 + t * Ns] = v[t];
}

// Unpack the packed-real transform: Z = FFT_M(y[0::2] + i y[1::2]).  For 1 <= j <= M/2, with
// (a,b) = Z_j, (c,d) = Z_{M-j} and (co,si) = (cos, sin)(2 pi j / N), N = 2M:
//   Im F_j = P - Q,  Im F_{M-j} = -P - Q,  P = (b-d)/2,  Q = si (b+d)/2 + co (a-c)/2.
HMG_HD void unpack_imag_pair(cplx zj, cplx zmj, double co, double si, double& imFj, double& imFmj) {
    const double P = 0.5 * (zj.y - zmj.y);
    const double Q = si * (0.5 * (zj.y + zmj.y)) + co * (0.5 * (zj.x - zmj.x));
    imFj = P - Q;
    imFmj = -P - Q;
}

// ---- Long radial grids with short support: the pruned decomposition (hmvec/fft.py:56-94 with nxs = 30000 / 40000,
// the lengths the reference's own callers use: examples/lensing_baryons.py:27, hmvec/params.py:59-60).
// The packed row z_p (M = nxs/2 complex samples) does not fit LDS, but the profile is cut at cmax << xmax: z_p = 0
// for p >= P0.  With LP >= P0, M = R LP:
//     Z[r + R q] = sum_{p < LP} (z_p W_M^{r p}) W_LP^{q p},        r < R, q < LP
// - R transforms of length LP of the row multiplied by W_M^{rp} - and the unpack step of the packed-real
// transform pairs mode j = r + R q with M - j, which sits in residue (R - r) mod R at quotient LP - 1 - q
// (LP - q for r == 0).  Residues r and R - r are therefore transformed together (two LDS buffers) and unpacked
// on the spot; only the quotients that reach a needed mode j <= jn are kept.
// per-mode constants of the unpack step, one 32-byte load: the rotation of the packed-real transform and the
// reciprocals that turn Im F_j into u_j = -Im F_j step / (kt_j mnorm) with kt_j = j kt_1 (the modes of an FFT
// sit on a uniform grid: np.fft.rfftfreq) - a table value and one product instead of a reciprocal per mode.
struct alignas(32) UnpackTw {
    double co, si, rj, rmj;      // (cos, sin)(2 pi j / N), 1/j, 1/(M-j)
};
struct PrunedPair {
    int j;        // the mode, 1 <= j; its mirror is M - j
    int qp;       // quotient of the mirror in the partner residue's transform
};
HMG_HD PrunedPair pruned_pair(int R, int LP, int s, int q) {
    return PrunedPair{s + R * q, s == 0 ? LP - q : LP - 1 - q};
}
// One pass (index PS of the compile-time plan of length LP) over up to two buffers laid out back to back:
// butterfly jj of the batch is butterfly jj mod nb of buffer jj / nb.
template <int LP, int PS>
struct SubPass {
    static constexpr FftPlanDev P = fft_plan_c(LP);
    static constexpr int R = P.radix[PS], Ns = P.ns[PS], tws = P.twstep[PS], nb = LP / R;
    static constexpr bool SMALL = nb <= 1024 && Ns <= 1024;
    static constexpr unsigned mg = SMALL ? small_magic((unsigned)Ns) : P.magic[PS];
    static constexpr bool last = PS == P.npass - 1;
    static constexpr int twoff = pass_tw_offset(P, PS);      // this pass's slice of the per-pass twiddle table
};
// twP: the per-pass twiddle table of the length-LP plan (pass_tw_table)
template <int LP, int PS>
HMG_HD void sub_pass_load(const cplx* buf, const cplx* twP, int jj, cplx* v) {
    using S = SubPass<LP, PS>;
    const int b = jj >= S::nb ? 1 : 0, j = jj - b * S::nb;
    pass_load<S::R, S::SMALL>(buf + b * LP, twP + S::twoff, LP, S::Ns, 1, S::mg, j, v);
}
// the same with the twiddle held by the caller: sub_pass_twiddle once, sub_pass_load_w per transform
template <int LP, int PS>
HMG_HD cplx sub_pass_twiddle(const cplx* twP, int jj) {
    using S = SubPass<LP, PS>;
    const int j = jj >= S::nb ? jj - S::nb : jj;
    return twP[S::twoff + pass_twiddle_index<S::R, S::SMALL>(S::Ns, 1, S::mg, j)];
}
template <int LP, int PS>
HMG_HD void sub_pass_load_w(const cplx* buf, int jj, cplx w1, cplx* v) {
    using S = SubPass<LP, PS>;
    const int b = jj >= S::nb ? 1 : 0, j = jj - b * S::nb;
    pass_load_w<S::R, S::SMALL>(buf + b * LP, LP, j, w1, v);
}
template <int LP, int PS>
HMG_HD void sub_pass_store(cplx* buf, int jj, cplx* v) {
    using S = SubPass<LP, PS>;
    const int b = jj >= S::nb ? 1 : 0, j = jj - b * S::nb;
    pass_store<S::R, S::SMALL>(buf + b * LP, S::Ns, S::mg, j, v);
}
// butterfly jj of the last pass is needed when its buffer-local index is within `keep` of either end
// (outputs q <= keep and q >= LP - keep: see fused_pass); keep < 0 keeps everything
template <int LP, int PS>
HMG_HD bool sub_pass_active(int jj, int nbuf, int keep) {
    using S = SubPass<LP, PS>;
    if (jj >= nbuf * S::nb) return false;
    if (!S::last || keep < 0) return true;
    const int j = jj >= S::nb ? jj - S::nb : jj;
    return j <= keep || j >= S::nb - keep;
}

// Tables and the scratch line of the decomposition are laid out BY RESIDUE, so that the threads of a wavefront - which
// walk consecutive quotients q of one residue s - touch consecutive elements (by mode j = s + R q they would be R
// elements apart: one cache line per lane):
//   twR[s LP + p]  = W_M^(s p)            the residue's twiddles on the samples           (residue_tw_table)
//   twNr[s QS + q] = unpack constants of mode j = s + R q <= M/2, QS = LP/2 + 1          (residue_unpack_table)
//   u[s LP + q]    = u_j, j = s + R q, 1 <= j < M                                         (pruned_u_index)
HMG_HD int pruned_qs(int LP) { return LP / 2 + 1; }
HMG_HD int pruned_u_index(int R, int LP, unsigned rmagic, int j) {
    const int q = (int)fast_div((unsigned)j, rmagic);      // rmagic = 2^32 / R + 1 (exact far beyond any M)
    return (j - q * R) * LP + q;
}
inline std::vector<cplx> residue_tw_table(int M, int LP) {
    const int R = M / LP;
    std::vector<cplx> t((size_t)R * LP);
    const long double twopi = 6.283185307179586476925286766559L;
    for (int s = 0; s < R; ++s)
        for (int p = 0; p < LP; ++p) {
            const long double a = twopi * (long double)(((long long)s * p) % M) / (long double)M;
            t[(size_t)s * LP + p] = cplx{(double)cosl(a), (double)-sinl(a)};
        }
    return t;
}
// Unpack step of one residue of a transformed pair: thread `tid` of `nthreads` walks the quotients q = tid,
// tid + nthreads, ... of residue s (its transform in buffer ob, the partner residue's in buffer pb), forms
// Im F_j and Im F_{M-j} of every needed pair (j <= jn, or the mirror M - j <= jn) and stores
// u_j = Im F_j * sc / j at its place in the residue-major line.  Modes above M/2 are reached as mirrors from the
// partner residue: M - j sits in residue (R - s) mod R at quotient pr.qp.
HMG_HD void pruned_unpack(const cplx* buf, int LP, int R, int M, int s, int ob, int pb, int jn, const UnpackTw* twNr,
                          double sc, double* u, int tid, int nthreads) {
    const int half = M / 2;
    const bool hi_any = jn >= M - half;             // does any mirror M - j with j <= M/2 lie within jn at all
    const UnpackTw* tws = twNr + (size_t)s * pruned_qs(LP);
    double* us = u + (size_t)s * LP;
    double* um = u + (size_t)(s == 0 ? 0 : R - s) * LP;
    for (int q = tid + (s == 0 ? 1 : 0);; q += nthreads) {
        const PrunedPair pr = pruned_pair(R, LP, s, q);
        if (pr.j > half) break;
        const bool lo = pr.j <= jn, hi = M - pr.j <= jn;
        if (!lo && !hi) {
            if (!hi_any) break;                     // j grows with q: nothing further is needed
            continue;
        }
        const cplx zj = buf[ob * LP + q], zmj = buf[pb * LP + pr.qp];
        const UnpackTw w = tws[q];
        double fa, fb;
        unpack_imag_pair(zj, zmj, w.co, w.si, fa, fb);
        us[q] = fa * sc * w.rj;
        if (hi && M - pr.j >= 1) um[pr.qp] = fb * sc * w.rmj;
    }
}
inline std::vector<UnpackTw> residue_unpack_table(int M, int LP) {
    const int R = M / LP, QS = pruned_qs(LP), n = 2 * M;
    std::vector<UnpackTw> t((size_t)R * QS, UnpackTw{1.0, 0.0, 0.0, 0.0});
    const long double twopi = 6.283185307179586476925286766559L;
    for (int s = 0; s < R; ++s)
        for (int q = 0; q < QS; ++q) {
            const int j = s + R * q;
            if (j > M / 2) break;
            t[(size_t)s * QS + q] = UnpackTw{(double)cosl(twopi * j / n), (double)sinl(twopi * j / n), j ? 1.0 / j : 0.0, 1.0 / (M - j)};
        }
    return t;
}
// Which residues a group g = 0 .. R/2 transforms: g and R - g (one buffer for the self-paired g = 0 and 2g = R),
// and whether a row that needs the modes j <= jn needs the group at all.
HMG_HD int pruned_group_partner(int R, int g) { return (g == 0 || 2 * g == R) ? -1 : R - g; }
HMG_HD bool pruned_group_needed(int R, int M, int g, int jn) {
    return jn >= g || jn >= M - M / 2;              // the smallest mode of the group is g (R - g > g); mirrors: all groups
}
// last-pass pruning of a group's transforms: outputs q <= Q and q >= LP - 1 - Q with Q = jn / R + 1 (keep = Q + 1 in
// fused_pass's convention), nothing pruned when mirrors are needed or the band is not narrow
HMG_HD int pruned_keep(int R, int M, int nb_last, int jn) {
    if (jn >= M - M / 2) return -1;
    const int keep = jn / R + 2;
    return 2 * keep + 2 < nb_last ? keep : -1;
}

// ---- Rows that need FEW modes of a long grid: the chirp transform.  The pruned decomposition above costs R
// transforms of length LP whatever the row needs; most rows of a halo-model grid need a few hundred of the M modes
// (jn = max(ks)/k_lo + 3; median 394 of 15000 on the Config-3 grid at xmax = 50).  With j p = (j^2 + p^2 - (j-p)^2)/2,
//     Z[j] = ch(j) sum_{p < P0} (z_p ch(p)) conj(ch(j - p)),      ch(n) = exp(-i pi n^2 / M),
// a linear convolution of the P0 chirped samples with the chirp, i.e. ONE forward transform of length
// Lc >= P0 + 2 Jw, a pointwise product with the (tabulated) transform of the chirp window and one more transform:
// every mode |j| <= Jw = (Lc - P0)/2 at the cost of two length-Lc transforms instead of R length-LP ones.
// The chirp window is laid out circularly - g(n) = conj(ch(n)) at index n for 0 <= n <= Jw and at Lc + n for
// -(Jw + P0 - 1) <= n < 0 - so that c_j lands at index j mod Lc; with the inverse transform taken as a forward one
// read backwards (IFFT(X)[i] = FFT(X)[(Lc - i) mod Lc] / Lc), Z[j] = ch(j) Y[Lc - j] and Z[-j] = ch(j) Y[j]:
// the needed outputs sit at the two ends of Y, where the last pass's pruning (keep) applies.
// chirp angles are reduced exactly in integers (n^2 mod 2M) before any floating point.
// Rows beyond the central window take up to `nwin` further PAIRS of one-sided windows of Kp = Lc - P0 + 1 modes each,
//     + window w: j in [j0, j0 + Kp),  j0 = Jw + 1 + (w-1) Kp,  window g(j0 - (P0-1) + m):  Z[j]  = ch(j) Y+[Kp - (j - j0)]
//     - window w: the mirrored modes -j,                  window g(j0 + Kp - 1 + P0 - 1 - m): Z[-j] = ch(j) Y-[1 + (j - j0)]
// (g is even).  Each costs one more length-Lc transform of the SAME forward transform A times another tabulated window
// transform, so a row that needs jn <= Jw + nwin Kp modes costs 2 + 2 ceil((jn - Jw)/Kp) transforms.
struct ChirpTables {
    int M = 0, Lc = 0, P0 = 0, Jw = 0, Kp = 0, nwin = 0;
    std::vector<cplx> chP;    // ch(p), p < Lc/2 (the samples a thread can own; zero-support entries are harmless)
    std::vector<cplx> chJ;    // ch(j), 0 <= j <= Jw + nwin Kp
    std::vector<cplx> Bw;     // FFT_Lc(chirp window) / Lc: central window, then (+1, -1, +2, -2, ...), Lc entries each
};
inline cplx chirp_value(long long n, int M) {
    const long long r = (n * n) % (2LL * M);
    const long double ang = 3.14159265358979323846264338327950288L * (long double)r / (long double)M;
    return cplx{(double)cosl(ang), (double)-sinl(ang)};
}
// forward DFT in long double, recursive by the smallest prime factor (table plans only: a few per context)
struct ldc { long double re, im; };
inline void ld_dft(const ldc* x, int n, int stride, ldc* out, const ldc* tw, int N) {
    if (n == 1) { out[0] = x[0]; return; }
    int p = 2;
    while (n % p) ++p;
    const int m = n / p;
    std::vector<ldc> sub((size_t)n);
    for (int r = 0; r < p; ++r) ld_dft(x + (size_t)r * stride, m, stride * p, sub.data() + (size_t)r * m, tw, N);
    const int step = N / n;                              // W_n = tw[step]
    for (int q = 0; q < p; ++q)
        for (int k = 0; k < m; ++k) {
            long double sr = 0.0L, si = 0.0L;
            const long long kk = (long long)k + (long long)q * m;
            for (int r = 0; r < p; ++r) {
                const ldc w = tw[(size_t)((r * kk) % n) * step];
                const ldc y = sub[(size_t)r * m + k];
                sr += y.re * w.re - y.im * w.im;
                si += y.re * w.im + y.im * w.re;
            }
            out[kk] = ldc{sr, si};
        }
}
inline ChirpTables chirp_make_tables(int M, int Lc, int P0, int nwin = 0) {
    ChirpTables T;
    T.M = M; T.Lc = Lc; T.P0 = P0;
    T.Jw = (Lc - P0) / 2;
    T.Kp = Lc - P0 + 1;
    if (T.Jw > M / 2 - 1) T.Jw = M / 2 - 1;
    while (nwin > 0 && T.Jw + nwin * T.Kp > M / 2 - 1) --nwin;      // mirrors are never needed on this route
    T.nwin = nwin;
    T.chP.resize(Lc / 2);
    for (int p = 0; p < Lc / 2; ++p) T.chP[p] = chirp_value(p, M);
    const int jmax = T.Jw + nwin * T.Kp;
    T.chJ.resize(jmax + 1);
    for (int j = 0; j <= jmax; ++j) T.chJ[j] = chirp_value(j, M);
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<ldc> tw((size_t)Lc), b((size_t)Lc), B((size_t)Lc);
    for (int t = 0; t < Lc; ++t) tw[t] = ldc{cosl(2 * pi * t / Lc), -sinl(2 * pi * t / Lc)};
    auto g = [&](long long n) {                          // conj(ch(n)), angle reduced in integers
        const long long r = (n * n) % (2LL * M);
        const long double ang = pi * (long double)r / (long double)M;
        return ldc{cosl(ang), sinl(ang)};
    };
    auto emit = [&]() {
        ld_dft(b.data(), Lc, 1, B.data(), tw.data(), Lc);
        for (int k = 0; k < Lc; ++k) T.Bw.push_back(cplx{(double)(B[k].re / Lc), (double)(B[k].im / Lc)});
    };
    // central window, laid out circularly: g(n) at n for 0 <= n <= Jw, at Lc + n for -(Jw + P0 - 1) <= n < 0
    for (int m = 0; m < Lc; ++m) b[m] = ldc{0.0L, 0.0L};
    for (int n = 0; n <= T.Jw; ++n) b[n] = g(n);
    for (int n = 1; n <= T.Jw + P0 - 1; ++n) b[Lc - n] = g(-n);
    emit();
    for (int w = 1; w <= nwin; ++w) {
        const long long j0 = T.Jw + 1 + (long long)(w - 1) * T.Kp;
        for (int m = 0; m < Lc; ++m) b[m] = g(j0 - (P0 - 1) + m);
        emit();
        for (int m = 0; m < Lc; ++m) b[m] = g(j0 + T.Kp - 1 + (P0 - 1) - m);
        emit();
    }
    return T;
}
// first pass (radix 4, sub-transform size 1) of the forward transform of the chirped row: the thread owns samples
// p = j and j + Lc/4 (the row is zero from Lc/2 on), i.e. inputs (a0, a1, 0, 0) of butterfly j
HMG_HD void chirp_first_pass(cplx z0, cplx z1, cplx c0, cplx c1, cplx* v) {
    const cplx a0 = cmul(z0, c0), a1 = cmul(z1, c1);
    v[0] = cadd(a0, a1);
    v[1] = cadd(a0, cmul_mi(a1));
    v[2] = csub(a0, a1);
    v[3] = csub(a0, cmul_mi(a1));
}
// one-sided windows: Z[j] from the + transform, Z[-j] from the - transform (j0 <= j < j0 + Kp)
HMG_HD cplx chirp_plus(const cplx* Y, int Kp, int j0, int j, cplx chj) { return cmul(chj, Y[Kp - (j - j0)]); }
HMG_HD cplx chirp_minus(const cplx* Y, int j0, int j, cplx chj) { return cmul(chj, Y[1 + (j - j0)]); }
// mode j (1 <= j <= Jw) from the transformed product Y: Z[j] = ch(j) Y[Lc - j], Z[M - j] = Z[-j] = ch(j) Y[j]
HMG_HD double chirp_unpack(const cplx* Y, int Lc, int j, cplx chj, const UnpackTw& w) {
    const cplx zj = cmul(chj, Y[Lc - j]), zmj = cmul(chj, Y[j]);
    double fa, fb;
    unpack_imag_pair(zj, zmj, w.co, w.si, fa, fb);
    return fa;
}

// ---- Rows whose support does NOT prune but which need few modes: the narrow-band route (the tSZ notebook's
// add_battaglia_pres_profile(xmax=2, nxs=30000): the profile fills half the grid and more, while the halo scale R_200c
// and xmax = 2 put every needed mode below j ~ 250 of 15000).  Decimation in time with the OUTPUT pruned: with
// p = p1 + D p2, M = D LB,
//     Z[j] = sum_{p1 < D} W_M^{j p1} Y_p1[j mod LB],      Y_p1 = FFT_LB(z[p1 + D p2], p2 < LB),
// - D transforms of length LB >= 2 jn + 2 of the decimated row, of which only the outputs at the two ends are needed
// (the last pass's pruning), accumulated per needed mode with a running twiddle W_M^{j p1} = (W_M^j)^{p1}.  Nothing
// of length M is ever held: one length-LB buffer in LDS, one accumulator per mode in registers.
// The owner of accumulator slot t (0 <= t < 2 jn + 1) holds mode j = t - jn and reads Y at j mod LB:
HMG_HD int band_mode(int t, int jn) { return t - jn; }
HMG_HD int band_index(int j, int LB) { return j < 0 ? LB + j : j; }

}  // namespace hmg
