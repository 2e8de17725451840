// Kernel-boundary cost on this part: N dependent launches of (a) an empty kernel, (b) a kernel that reads and writes one
// cache line, timed on the stream with HIP events -- plain launches and the same chain replayed as a HIP graph.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/launch_floor.hip -o build/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] + 1.0f; }
__global__ void k_wide(float* p) { p[blockIdx.x * 256 + threadIdx.x] += 1.0f; }       // 256 blocks: one per CU

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int N = 2000;
    float* d;
    CK(hipMalloc(&d, 256 * 256 * sizeof(float)));
    CK(hipMemset(d, 0, 256 * 256 * sizeof(float)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        auto launch = [&]() {
            if (variant == 0) k_empty<<<1, 64, 0, s>>>();
            else if (variant == 1) k_touch<<<1, 64, 0, s>>>(d);
            else k_wide<<<256, 256, 0, s>>>(d);
        };
        for (int i = 0; i < 100; ++i) launch();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float plain = 1e3f * ms / N;
        // the same chain as a graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) launch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us per launch (stream), %.2f us per node (graph)\n",
               variant == 0 ? "empty kernel" : variant == 1 ? "one-line read-modify-write" : "256 blocks x 256 threads rmw",
               plain, 1e3f * ms / N);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
