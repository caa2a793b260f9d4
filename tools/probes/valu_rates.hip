// Issue rate of the fp64 VALU instructions the hot kernels are made of (MI355X): cycles per wave64 instruction on one
// SIMD, measured with 8 wavefronts per SIMD and eight independent chains per lane so that latency is hidden.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048, CH = 8;
template <int OP>
__global__ __launch_bounds__(512) void rate(double* out, double seed, int sel) {
    double v[CH];
    for (int c = 0; c < CH; ++c) v[c] = seed + 1e-3 * (threadIdx.x + c);
    int acc = sel;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == 0) v[c] = __builtin_fma(v[c], 1.0000001, 1e-9);
            else if (OP == 1) v[c] = v[c] + 1.0000001;
            else if (OP == 2) v[c] = v[c] * 1.0000001;
            else if (OP == 3) v[c] = __builtin_rint(v[c] * 1.5);                 // v_mul + v_rndne
            else if (OP == 4) v[c] = __builtin_amdgcn_rcp(v[c]);
            else if (OP == 5) { acc += (int)v[c]; v[c] += 1.0; }                 // v_cvt_i32_f64 + v_add_f64 (+ v_add_u32)
            else if (OP == 6) v[c] = __builtin_amdgcn_ldexp(v[c], acc & 1);
            else if (OP == 7) v[c] = __builtin_amdgcn_frexp_mant(v[c]) + 1.0;    // v_frexp_mant + v_add
            else if (OP == 8) v[c] = (double)(acc + c) + v[c];                   // v_cvt_f64_i32 + v_add (+ int add)
            else if (OP == 9) v[c] = v[c] > 1.5 ? v[c] - 0.5 : v[c] + 0.25;      // cmp + cndmask x2 + 2 adds
            else if (OP == 10) v[c] = __builtin_fmin(v[c] * 1.0000001, 1.0e9);   // v_mul + v_min
            else if (OP == 11) v[c] = __builtin_amdgcn_rsq(v[c]) + 1.0;
            else if (OP == 12) v[c] = __builtin_sqrt(v[c]) + 1.0;
        }
    }
    double s = 0.0;
    for (int c = 0; c < CH; ++c) s += v[c];
    if (s == 1.2345e300) out[threadIdx.x] = s + acc;
}
template <int OP>
static int run(const char* name, int nvalu, double* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int w = 0; w < 200; ++w)      // ~50 ms of the same work first: the clock a sustained fp64 load actually gets
        hipLaunchKernelGGL(rate<OP>, dim3(256 * 4), dim3(512), 0, 0, out, 1.25, 0);
    for (int rep = 0; rep < 20; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate<OP>, dim3(256 * 4), dim3(512), 0, 0, out, 1.25, 0);   // 8 waves per SIMD, one round
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // cycles per wave-instruction group on a SIMD: time * 2.4e9 / (8 waves * ITER * CH)
    const double cyc = best * 1e-3 * 2.4e9 / (8.0 * ITER * CH);
    printf("%-44s %8.4f ms  %6.2f cycles per chain step (%d VALU instr. in it: %5.2f each at 2.4 GHz)\n", name, best, cyc, nvalu, cyc / nvalu);
    return 0;
}
int main() {
    double* out; CK(hipMalloc(&out, 4096));
    run<0>("v_fma_f64", 1, out); run<1>("v_add_f64", 1, out); run<2>("v_mul_f64", 1, out);
    run<3>("v_mul_f64 + v_rndne_f64", 2, out); run<4>("v_rcp_f64", 1, out);
    run<5>("v_cvt_i32_f64 + v_add_f64 + v_add_u32", 3, out); run<6>("v_ldexp_f64 (+ v_and_b32)", 2, out);
    run<7>("v_frexp_mant_f64 + v_add_f64", 2, out); run<8>("v_cvt_f64_i32 + v_add_f64 + v_add_u32", 3, out);
    run<9>("v_cmp + 2 v_add_f64 + 2 v_cndmask_b32", 5, out); run<10>("v_mul_f64 + v_min_f64", 2, out);
    run<11>("v_rsq_f64 + v_add_f64", 2, out); run<12>("sqrt (library sequence) + v_add_f64", 2, out);
    return 0;
}
