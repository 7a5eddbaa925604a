// Launch-to-launch cost of a chain of dependent small kernels on one stream: plain launches vs one hipGraph.
// Build: hipcc --offload-arch=gfx950 -O2 -o launch_gap tools/micro/launch_gap.cpp   (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void bump(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}

int main() {
    const int n = 256 * 256, chain = 400;
    double* d = nullptr;
    CK(hipMalloc(&d, n * sizeof(double)));
    CK(hipMemset(d, 0, n * sizeof(double)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run_plain = [&]() { for (int i = 0; i < chain; ++i) bump<<<256, 256, 0, s>>>(d, n); };
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        run_plain();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("plain : %.2f us per kernel (chain of %d)\n", ms * 1e3 / chain, chain);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    run_plain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("graph : %.2f us per kernel (chain of %d)\n", ms * 1e3 / chain, chain);
    }
    return 0;
}
