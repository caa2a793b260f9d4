// Does the 32-KiB row stride of the (z,m,k) tensors (nk = 4096 doubles) cost the mass integrals bandwidth?  The access
// shape of hmg::power_batch_kernel (1-KiB segments of two tensors, 8 wavefronts of a block on interleaved mass rows) with
// the rows laid out at a stride of nk + pad doubles.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stride_probe.hip -o /tmp/stride_probe && /tmp/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(512) void shaped_read(const double* __restrict__ t0, const double* __restrict__ t1,
                                                   int nm, int ldk, double* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, z = blockIdx.y;
    const int k0 = (blockIdx.x * 64 + lane) * 2;
    double s = 0.0;
    for (int m = wv; m < nm; m += 8) {
        const size_t off = ((size_t)z * nm + m) * ldk + k0;
        const double2 a = *reinterpret_cast<const double2*>(t0 + off);
        const double2 b = *reinterpret_cast<const double2*>(t1 + off);
        s += a.x * b.y + a.y * b.x;
    }
    if (s == 1.2345e-300) out[0] = s;
}
__global__ __launch_bounds__(256) void row_write8(double* __restrict__ t, int nk, int ldk) {
    double* row = t + (size_t)blockIdx.x * ldk;
    const double base = 1.0 / (1.0 + blockIdx.x);
    for (int i = threadIdx.x; i < nk; i += 256) __builtin_nontemporal_store(base + 1e-6 * i, &row[i]);
}
int main() {
    const int nz = 32, nm = 512, nk = 4096, maxpad = 512;
    const size_t n = (size_t)nz * nm * (nk + maxpad);
    double *t0, *t1, *out;
    CK(hipMalloc(&t0, n * 8)); CK(hipMalloc(&t1, n * 8)); CK(hipMalloc(&out, 8));
    {
        std::vector<double> h(n);
        unsigned long long x = 88172645463325252ull;
        for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (double)(x >> 11) / 9007199254740992.0 + 1e-9; }
        CK(hipMemcpy(t0, h.data(), n * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < n; ++i) h[i] = 1.0 - 0.5 * h[i];
        CK(hipMemcpy(t1, h.data(), n * 8, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * nz * nm * (double)nk * 8;
    const int pads[] = {0, 2, 16, 32, 64, 96, 128, 256, 512};
    for (int w = 0; w < 60; ++w) hipLaunchKernelGGL(shaped_read, dim3(nk / 128, nz), dim3(512), 0, 0, t0, t1, nm, nk, out);
    for (int pad : pads) {
        float best = 1e30f, bw = 1e30f;
        for (int rep = 0; rep < 14; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(shaped_read, dim3(nk / 128, nz), dim3(512), 0, 0, t0, t1, nm, nk + pad, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < best) best = ms;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(row_write8, dim3(nz * nm), dim3(256), 0, 0, t0, nk, nk + pad);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < bw) bw = ms;
        }
        printf("row stride nk + %3d doubles: shaped read of two tensors %.4f ms = %.0f GB/s   row write of one %.4f ms = %.0f GB/s\n",
               pad, best, bytes / (best * 1e-3) / 1e9, bw, bytes / 2 / (bw * 1e-3) / 1e9);
    }
    return 0;
}
