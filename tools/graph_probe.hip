// Launch floor of a chain of small dependent kernels: stream launches against a captured hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_probe.hip -o tools/graph_probe && tools/graph_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void tiny(unsigned *p, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v; }

int main() {
    unsigned *d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int N = 10, IT = 200;
    for (int grid : {1, 512}) {
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, 1u);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int it = 0; it < IT; ++it)
            for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, 1u);
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("grid %4d  stream launches: %.2f us per kernel\n", grid, ms * 1000.0f / (N * IT));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d, 1u);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(a, s));
        for (int it = 0; it < IT; ++it) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("grid %4d  graph launches:  %.2f us per kernel\n", grid, ms * 1000.0f / (N * IT));
    }
    return 0;
}
