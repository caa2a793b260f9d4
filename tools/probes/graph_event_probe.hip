// Probe (run on the GPU box): do events recorded by HIP-graph event-record nodes carry timestamps
// that hipEventElapsedTime accepts, and which capture forms work on this runtime?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); printf("%-62s -> %s\n", #x, hipGetErrorString(e)); if (e != hipSuccess) (void)hipGetLastError(); } while (0)
__global__ void spin(double* p, int n) { double a = p[threadIdx.x]; for (int i = 0; i < n; ++i) a = a * 1.0000001 + 1e-9; p[threadIdx.x] = a; }
int main() {
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    double* d; CK(hipMalloc(&d, 1024 * 8)); CK(hipMemset(d, 0, 1024 * 8));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ef)); CK(hipEventCreate(&ej));
    // 1. plain capture with event records inside
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 200000);
    CK(hipEventRecord(e1, s));
    // fork to s2 and join back
    CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
    hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s2, d + 512, 100000);
    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s, ej, 0));
    hipGraph_t g; CK(hipStreamEndCapture(s, &g));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("graph nodes: %zu\n", nn);
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 3; ++it) {
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float ms = -1; hipError_t e = hipEventElapsedTime(&ms, e0, e1);
        printf("replay %d: hipEventElapsedTime -> %s, %.4f ms\n", it, hipGetErrorString(e), ms); (void)hipGetLastError();
        printf("  hipEventQuery(e1) -> %s\n", hipGetErrorString(hipEventQuery(e1))); (void)hipGetLastError();
    }
    // 2. eager records around a graph launch (always valid): timing of the whole replay
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms = -1; CK(hipEventElapsedTime(&ms, e0, e1)); printf("eager bracket around replay: %.4f ms\n", ms);
    return 0;
}
