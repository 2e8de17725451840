// Does hipGraphExecUpdate while earlier launches of the same executable graph are still in flight keep those launches'
// arguments?  Each repetition r adds r (exactly representable) NK times; the final sum tells.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Args { double* p; int n; double add; long pad[20]; };
__global__ void k_add(Args a) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < a.n) a.p[i] += a.add; }
// a slow kernel in front keeps the queue full so that updates run far ahead of execution
__global__ void k_spin(double* p, int iters) { double x = p[0]; for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9; if (x == -1.0) p[0] = x; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int NK = 9, REP = 400, n = 256 * 256;
    double* d;
    CK(hipMalloc(&d, n * sizeof(double)));
    CK(hipMemset(d, 0, n * sizeof(double)));
    hipStream_t s, cap;
    CK(hipStreamCreate(&s));
    CK(hipStreamCreate(&cap));
    auto chain = [&](hipStream_t st, int r) { for (int k = 0; k < NK; ++k) { Args a{d, n, (double)r, {}}; k_add<<<256, 256, 0, st>>>(a); } };
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    chain(cap, 0);
    CK(hipStreamEndCapture(cap, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    k_spin<<<1, 1, 0, s>>>(d + 1, 20000000);                   // ~tens of ms: everything below queues up behind it
    int fails = 0;
    for (int r = 1; r <= REP; ++r) {
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));   // captured on an internal stream ...
        chain(cap, r);
        CK(hipStreamEndCapture(cap, &g));
        hipGraphExecUpdateResult res; hipGraphNode_t bad;
        if (hipGraphExecUpdate(ge, g, &bad, &res) != hipSuccess) ++fails;
        CK(hipGraphDestroy(g));
        CK(hipGraphLaunch(ge, s));                                          // ... launched on the caller's stream
    }
    CK(hipStreamSynchronize(s));
    double h[2];
    CK(hipMemcpy(h, d + 100, sizeof(h), hipMemcpyDeviceToHost));
    const double want = (double)NK * REP * (REP + 1) / 2;
    printf("sum %.1f want %.1f %s, update failures %d\n", h[0], want, h[0] == want ? "OK" : "MISMATCH", fails);
    return 0;
}
