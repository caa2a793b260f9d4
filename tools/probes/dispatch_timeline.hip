// When do the wavefronts of a thin launch start?  N workgroups of WAVES wavefronts, each wavefront spins a fixed chain
// of dependent fp64 FMAs (WORK of them, ~4 cycles each when alone on its SIMD) and records wall_clock64() at entry and
// exit.  Prints the kernel's event time and the distribution of start and end times relative to the first start.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probes/dispatch_timeline.hip -o /tmp/dt && /tmp/dt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int REGS>
__global__ void spin(unsigned long long* t, int work, double seed, double* sink) {
    const unsigned long long t0 = wall_clock64();
    double v[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) v[i] = seed + i + threadIdx.x;
    for (int it = 0; it < work; ++it) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) v[i] = __builtin_fma(v[i], 1.0000001, 1e-9);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += v[i];
    if (s == 1.2345e300) sink[0] = s;
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        t[2 * w] = t0; t[2 * w + 1] = t1;
    }
}
template <int REGS>
static int run(const char* name, int nwg, int waves, int work) {
    const size_t nw = (size_t)nwg * waves;
    unsigned long long* d; double* sink;
    CK(hipMalloc(&d, nw * 16)); CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(spin<REGS>, dim3(nwg), dim3(64 * waves), 0, 0, d, work, 1.0, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * nw);
    CK(hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost));
    unsigned long long first = ~0ull;
    for (size_t w = 0; w < nw; ++w) first = std::min(first, h[2 * w]);
    std::vector<double> st(nw), en(nw), life(nw);
    for (size_t w = 0; w < nw; ++w) { st[w] = (h[2*w] - first) * 0.01; en[w] = (h[2*w+1] - first) * 0.01; life[w] = en[w] - st[w]; }   // 100 MHz -> us
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end()); std::sort(life.begin(), life.end());
    auto q = [&](std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
    printf("%-34s wg=%5d x %d waves, work=%5d: event %7.1f us | start p50 %5.1f p90 %5.1f p99 %5.1f max %5.1f | end p50 %5.1f max %5.1f | life p50 %5.1f max %5.1f\n",
           name, nwg, waves, work, ms * 1e3, q(st, .5), q(st, .9), q(st, .99), st.back(), q(en, .5), en.back(), q(life, .5), life.back());
    CK(hipFree(d)); CK(hipFree(sink));
    return 0;
}
int main() {
    // ~0.7 us of work alone on a SIMD per 100 iterations of 4 chains (4 x 4 cycles x 100 / 2.3 GHz)
    run<4>("4 regs (8 VGPRs)", 4096, 1, 600); run<4>("4 regs", 1024, 4, 600); run<4>("4 regs", 4096, 1, 100);
    run<56>("56 regs (112+ VGPRs: 4 waves/SIMD)", 4096, 1, 43); run<56>("56 regs", 1024, 4, 43);
    run<56>("56 regs", 3072, 1, 43); run<56>("56 regs", 2048, 1, 43); run<56>("56 regs", 8192, 1, 43);
    run<4>("4 regs, 2048 x 4 waves", 2048, 4, 600); run<4>("4 regs, 2048 x 8 waves", 2048, 8, 300);
    return 0;
}
