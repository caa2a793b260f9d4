// Interface between hmgrid.hip (contexts, plans, the C ABI) and longgrid.hip (the long-grid row kernels, compiled as
// a translation unit of their own).  Host functions return a hipError_t as int.
#pragma once
#include "rowdev.hpp"

namespace hmg {

struct PrunedArgs {
    FusedArgs F;          // the row description (F.plan is not used)
    int M, R;             // packed length nxs/2 = R * LP
    const cplx* twB;      // exp(-2 pi i t / M), t < M (the narrow-band route's mode twiddles)
    const cplx* twR;      // pruned route: the residues' twiddles on the samples, [s LP + p] = W_M^(s p) (ldsfft.hpp)
    const UnpackTw* twNr; // pruned route: unpack constants by residue, [s (LP/2 + 1) + q] for mode s + R q
    unsigned rmagic;      // 2^32 / R + 1: j / R of the residue-major scratch line
    const cplx* twL;      // per-pass twiddle table of the sub-transform plan (length LP; the band route: LB)
    double* u;            // [rows of this launch][M]: u_j, j = s + R q, at [s LP + q] (pruned_u_index)
    int* fault;           // set when a row's support turns out longer than LP (stale support bound)
    int row0;             // first row of this launch
    // chirp route for rows that need few modes (ldsfft.hpp; nullptr: every row takes the decomposition)
    const cplx* chP;      // ch(p), p < LP
    const cplx* chJ;      // ch(j), j <= Jw
    const cplx* Bw;       // transform of the chirp window / Lc, Lc = 2 LP
    const cplx* twC;      // per-pass twiddle table of the length-Lc plan
    int Jw, p0;           // modes |j| <= Jw are in the window; it was built for supports of <= p0 packed samples
};

#ifndef HMG_LONG_NT
#define HMG_LONG_NT 512
#endif
constexpr int LONG_NT = HMG_LONG_NT;   // threads per row workgroup of the pruned long-grid kernel

// sub-transform lengths LP that are compiled in
bool pruned_lp_compiled(int LP);
// rows [0, rows) of G in launches of at most rows_per_launch rows (the scratch line block G.u holds that many)
int launch_pruned(hipStream_t stream, int LP, PrunedArgs G, int rows, size_t rows_per_launch);
// d_out[0] = max over rows of the packed samples that can be non-zero, d_out[1] = max over rows of the needed modes jn
// (rss == nullptr: left 0); both must be zero before
int launch_profile_support(hipStream_t stream, int rows, int nxs, const double* xs, const double* cmax, const double* rss,
                           const double* zs, int nm, const double* kts, const double* ks, int nk, int* d_out);
// the narrow-band route: transform lengths LB that are compiled in; G.R is D = M / LB, G.twL the table of length LB
bool band_lb_compiled(int LB);
int launch_band(hipStream_t stream, int LB, PrunedArgs G, int rows, int jnmax /* bound on the rows' needed modes */);

}  // namespace hmg
