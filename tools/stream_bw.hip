// Diagnostic: practical HBM READ bandwidth on this MI355X for (a) a flat grid-stride stream
// and (b) the access shape of hmg::power_batch_kernel (64 lanes x 16 B per row chunk, 8
// wavefronts per block on interleaved rows, two tensors), with no arithmetic beyond a sum.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/stream_bw.hip -o /tmp/stream_bw && /tmp/stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void flat_read(const double2* __restrict__ a, size_t n, double* out) {
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 1.2345e-300) out[0] = s;
}

// grid (nk/128, nz), block 512: wave w reads rows m = w, w+8, ... of its z, chunk blockIdx.x
// write-side reference points: the producers of the tensors (analytic NFW rows, fused profile rows) store one
// (nk) row per workgroup, 8 B per lane, non-temporal; values differ per element (an all-equal fill lets the chip
// clock higher)
__global__ __launch_bounds__(256) void row_write8(double* __restrict__ t, int nk) {
    double* row = t + (size_t)blockIdx.x * nk;
    const double base = 1.0 / (1.0 + blockIdx.x);
    for (int i = threadIdx.x; i < nk; i += 256) __builtin_nontemporal_store(base + 1e-6 * i, &row[i]);
}
typedef double v2d_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void row_write16(double* __restrict__ t, int nk) {
    v2d_t* row = reinterpret_cast<v2d_t*>(t + (size_t)blockIdx.x * nk);
    const double base = 1.0 / (1.0 + blockIdx.x);
    for (int i = threadIdx.x; i < nk / 2; i += 256) {
        const v2d_t v = {base + 2e-6 * i, base + 2e-6 * i + 1e-6};
        __builtin_nontemporal_store(v, &row[i]);
    }
}
__global__ __launch_bounds__(256) void row_write8_plain(double* __restrict__ t, int nk) {
    double* row = t + (size_t)blockIdx.x * nk;
    const double base = 1.0 / (1.0 + blockIdx.x);
    for (int i = threadIdx.x; i < nk; i += 256) row[i] = base + 1e-6 * i;
}

__global__ __launch_bounds__(512) void shaped_read(const double* __restrict__ t0, const double* __restrict__ t1,
                                                   int nm, int nk, double* out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, z = blockIdx.y;
    const int k0 = (blockIdx.x * 64 + lane) * 2;
    double s = 0.0;
    for (int m = wv; m < nm; m += 8) {
        const size_t off = ((size_t)z * nm + m) * nk + k0;
        const double2 a = *reinterpret_cast<const double2*>(t0 + off);
        const double2 b = *reinterpret_cast<const double2*>(t1 + off);
        s += a.x * b.y + a.y * b.x;
    }
    if (s == 1.2345e-300) out[0] = s;
}

// same shape with WORK dependent fp64 FMAs per loaded element pair (stand-in for the tracer forms)
template <int WORK>
__global__ __launch_bounds__(512) void shaped_read_work(const double* __restrict__ t0, const double* __restrict__ t1,
                                                        int nm, int nk, double* out) {
    extern __shared__ double pad[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, z = blockIdx.y;
    const int k0 = (blockIdx.x * 64 + lane) * 2;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = wv; m < nm; m += 8) {
        const size_t off = ((size_t)z * nm + m) * nk + k0;
        const double2 a = *reinterpret_cast<const double2*>(t0 + off);
        const double2 b = *reinterpret_cast<const double2*>(t1 + off);
#pragma unroll
        for (int w = 0; w < WORK; ++w) acc[w & 7] = fma(a.x + w, b.y, fma(a.y, b.x + w, acc[w & 7]));
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 1.2345e-300) out[0] = s + pad[0];
}

int main() {
    const int nz = 32, nm = 512, nk = 4096;
    const size_t n = (size_t)nz * nm * nk;
    double *t0, *t1, *out;
    CK(hipMalloc(&t0, n * 8)); CK(hipMalloc(&t1, n * 8)); CK(hipMalloc(&out, 8));
    {   // random, profile-like values in (0,1]: all-zero buffers let the chip clock higher (MI355X_MICROARCH.md, DVFS)
        std::vector<double> h(n);
        unsigned long long x = 88172645463325252ull;
        for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (double)(x >> 11) / 9007199254740992.0 + 1e-9; }
        CK(hipMemcpy(t0, h.data(), n * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < n; ++i) h[i] = 1.0 - 0.5 * h[i];
        CK(hipMemcpy(t1, h.data(), n * 8, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * n * 8;
    for (int variant = 0; variant < 11; ++variant) {
        float best = 1e30f;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0));
            if (variant == 0) {
                hipLaunchKernelGGL(flat_read, dim3(256 * 8), dim3(256), 0, 0, (const double2*)t0, n / 2, out);
                hipLaunchKernelGGL(flat_read, dim3(256 * 8), dim3(256), 0, 0, (const double2*)t1, n / 2, out);
            } else if (variant == 1) {
                hipLaunchKernelGGL(flat_read, dim3(256 * 32), dim3(256), 0, 0, (const double2*)t0, n / 2, out);
                hipLaunchKernelGGL(flat_read, dim3(256 * 32), dim3(256), 0, 0, (const double2*)t1, n / 2, out);
            } else if (variant == 2) {
                hipLaunchKernelGGL(shaped_read, dim3(nk / 128, nz), dim3(512), 0, 0, t0, t1, nm, nk, out);
            } else if (variant == 3) {
                hipLaunchKernelGGL(shaped_read_work<1>, dim3(nk / 128, nz), dim3(512), 60000, 0, t0, t1, nm, nk, out);
            } else if (variant == 4) {
                hipLaunchKernelGGL(shaped_read_work<16>, dim3(nk / 128, nz), dim3(512), 0, 0, t0, t1, nm, nk, out);
            } else if (variant == 5) {
                hipLaunchKernelGGL(shaped_read_work<32>, dim3(nk / 128, nz), dim3(512), 0, 0, t0, t1, nm, nk, out);
            } else if (variant == 6) {
                hipLaunchKernelGGL(shaped_read_work<32>, dim3(nk / 128, nz), dim3(512), 60000, 0, t0, t1, nm, nk, out);
            } else if (variant == 7) {
                hipLaunchKernelGGL(shaped_read_work<64>, dim3(nk / 128, nz), dim3(512), 0, 0, t0, t1, nm, nk, out);
            } else if (variant == 8) {
                hipLaunchKernelGGL(row_write8, dim3(nz * nm), dim3(256), 0, 0, t0, nk);
            } else if (variant == 9) {
                hipLaunchKernelGGL(row_write16, dim3(nz * nm), dim3(256), 0, 0, t0, nk);
            } else {
                hipLaunchKernelGGL(row_write8_plain, dim3(nz * nm), dim3(256), 0, 0, t0, nk);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < best) best = ms;
        }
        const char* name[11] = {"flat grid-stride read, 2048 blocks", "flat grid-stride read, 8192 blocks",
                               "power_batch access shape, no math", "shape + 2 FMA/iter, 2 blocks/CU (LDS pad)",
                               "shape + 32 FMA/iter", "shape + 64 FMA/iter", "shape + 64 FMA/iter, 2 blocks/CU",
                               "shape + 128 FMA/iter", "row write, 8 B/lane non-temporal (1 tensor)",
                               "row write, 16 B/lane non-temporal (1 tensor)", "row write, 8 B/lane plain (1 tensor)"};
        const double moved = variant >= 8 ? bytes / 2 : bytes;
        printf("%-44s %.4f ms  %.0f GB/s\n", name[variant], best, moved / (best * 1e-3) / 1e9);
    }
    return 0;
}
