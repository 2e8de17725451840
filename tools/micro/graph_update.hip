// Host cost of re-capturing a 9-kernel chain with new arguments and updating a cached executable graph with it
// (hipGraphExecUpdate), against 9 plain launches; and the GPU time of the chain either way.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Args { float* p; int n; float a; long pad[20]; };       // a fat by-value struct like the library's kernels take
__global__ void k_work(Args a) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < a.n) a.p[i] = a.p[i] * a.a + 1.0f; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int NK = 9, REP = 300, n = 256 * 256;
    float* d[2];
    CK(hipMalloc(&d[0], n * sizeof(float)));
    CK(hipMalloc(&d[1], n * sizeof(float)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto chain = [&](int r) {
        for (int k = 0; k < NK; ++k) { Args a{d[(r + k) & 1], n, 1.0f + 1e-6f * r, {}}; k_work<<<256, 256, 0, s>>>(a); }
    };
    for (int r = 0; r < 20; ++r) chain(r);
    CK(hipStreamSynchronize(s));
    // plain launches
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < REP; ++r) chain(r);
    CK(hipEventRecord(e1, s));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("plain : host %.1f us per chain, gpu %.1f us per chain\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / REP, 1e3 * ms / REP);
    // cached exec, re-capture + update per call
    hipGraph_t g;
    hipGraphExec_t ge = nullptr;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain(0);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    int fails = 0;
    t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < REP; ++r) {
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        chain(r);
        CK(hipStreamEndCapture(s, &g));
        hipGraphExecUpdateResult res;
        hipGraphNode_t bad;
        if (hipGraphExecUpdate(ge, g, &bad, &res) != hipSuccess) ++fails;
        CK(hipGraphDestroy(g));
        CK(hipGraphLaunch(ge, s));
    }
    CK(hipEventRecord(e1, s));
    t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph : host %.1f us per chain, gpu %.1f us per chain, update failures %d\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / REP, 1e3 * ms / REP, fails);
    float h[4];
    CK(hipMemcpy(h, d[0], sizeof(h), hipMemcpyDeviceToHost));
    printf("check %.3f\n", h[0]);
    return 0;
}
