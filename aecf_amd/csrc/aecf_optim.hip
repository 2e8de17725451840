// AdamW over a LIST of float32 tensors in one launch (the optimiser step of the example trainer, ref
// xrays/train_xrays_example.py:322-323, 376: torch.optim.AdamW(lr=1e-4, weight_decay=0.01)).
// torch's fused multi-tensor kernel hands a block a chunk of 65536 elements, so the ~1 M parameters of the example model run on
// ~30 blocks (40 us, tools/debug/adamw_time.py); here a block takes 1024 elements (16-byte accesses),
// the tensor table travels in the kernel arguments, and the per-tensor step counters live on the device (a captured step
// replays with a fresh count): every block reads the counters at its start, the LAST block to finish increments them.
//   p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//   p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps),   t = step + 1        (torch/optim/adamw.py, amsgrad off)
#include <math.h>

#include "aecf_kernels.h"

namespace aecf {

namespace {

constexpr int OPT_MAX_TENSORS = 24;
constexpr int OPT_BLOCK_ELEMS = 1024;
constexpr unsigned int OPT_SUBTICKETS = 64;          // + 1 top ticket: AECF_ADAMW_TICKET_WORDS uint32 per launch group

struct AdamArgs {
    float* p[OPT_MAX_TENSORS];
    const float* g[OPT_MAX_TENSORS];
    float* m[OPT_MAX_TENSORS];
    float* v[OPT_MAX_TENSORS];
    float* step[OPT_MAX_TENSORS];
    int64_t numel[OPT_MAX_TENSORS];
    unsigned int first_block[OPT_MAX_TENSORS + 1];
    unsigned int* ticket;
    int n;
    float lr, beta1, beta2, eps, weight_decay;
    float log_beta1, log_beta2;                       // natural logs, rounded from double on the host
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamArgs a) {
    int t = 0;
    while (t + 1 < a.n && blockIdx.x >= a.first_block[t + 1]) ++t;          // (<= 24 entries: linear)
    // 1 - beta^t = -expm1(t ln beta) in float32 (no cancellation for small t; torch's capturable path raises beta to a float32
    // step tensor as well); block-uniform, every lane forms it
    const float step = a.step[t][0] + 1.0f;
    const float step_size = a.lr / -expm1f(step * a.log_beta1), bc2_sqrt = sqrtf(-expm1f(step * a.log_beta2));
    const float decay = 1.0f - a.lr * a.weight_decay;
    const int64_t base = (int64_t)(blockIdx.x - a.first_block[t]) * OPT_BLOCK_ELEMS + 4 * threadIdx.x;
    const int64_t n = a.numel[t];
    float* p = a.p[t];
    const float* g = a.g[t];
    float* m = a.m[t];
    float* v = a.v[t];
    auto update = [&](float& pp, float gg, float& mm, float& vv) {
        pp *= decay;
        mm = a.beta1 * mm + (1.0f - a.beta1) * gg;
        vv = a.beta2 * vv + (1.0f - a.beta2) * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + a.eps;
        pp -= step_size * (mm / denom);
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (vec && base + 4 <= n) {
        f32x4 pv = *reinterpret_cast<f32x4*>(p + base), mv = *reinterpret_cast<f32x4*>(m + base);
        f32x4 vv = *reinterpret_cast<f32x4*>(v + base);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + base);
        float pa[4] = {pv[0], pv[1], pv[2], pv[3]}, ma[4] = {mv[0], mv[1], mv[2], mv[3]}, va[4] = {vv[0], vv[1], vv[2], vv[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) update(pa[i], gv[i], ma[i], va[i]);
        *reinterpret_cast<f32x4*>(p + base) = f32x4{pa[0], pa[1], pa[2], pa[3]};
        *reinterpret_cast<f32x4*>(m + base) = f32x4{ma[0], ma[1], ma[2], ma[3]};
        *reinterpret_cast<f32x4*>(v + base) = f32x4{va[0], va[1], va[2], va[3]};
    } else {
        for (int64_t i = base; i < base + 4 && i < n; ++i) update(p[i], g[i], m[i], v[i]);
    }
    // the last block to finish advances every tensor's counter (all other blocks have read theirs) and re-arms the tickets.
    // Two levels -- 64 sub-tickets by blockIdx % 64, whoever completes one takes a top ticket -- so that the thousands of blocks
    // of a large model do not queue on one address
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        // no fence: every lane's read of its step counter has been consumed by now (the barrier above), and that is all the last
        // block's write has to wait for.  A __threadfence() here made every block write back its XCD's L2 (the XCDs' L2s are
        // not coherent with each other): 0.74 TB/s at any size, 630 us for 16 M parameters
        int last = 0;
        const unsigned int sub = blockIdx.x % OPT_SUBTICKETS;
        const unsigned int expect = (gridDim.x - sub + OPT_SUBTICKETS - 1) / OPT_SUBTICKETS;
        if (atomicAdd(a.ticket + 1 + sub, 1u) == expect - 1) {
            a.ticket[1 + sub] = 0u;
            const unsigned int nsub = gridDim.x < OPT_SUBTICKETS ? gridDim.x : OPT_SUBTICKETS;
            if (atomicAdd(a.ticket, 1u) == nsub - 1) {
                a.ticket[0] = 0u;
                last = 1;
            }
        }
        is_last = last;
    }
    __syncthreads();
    // (one lane per tensor)
    if (is_last && (int)threadIdx.x < a.n) a.step[threadIdx.x][0] += 1.0f;
}

}  // namespace

// tensors in groups of OPT_MAX_TENSORS; ticket: 65 zero-initialised unsigned per group
void launch_adamw_multi(int n, float* const* p, const float* const* g, float* const* m, float* const* v, float* const* step,
                        const int64_t* numel, unsigned int* ticket, float lr, float beta1, float beta2, float eps, float weight_decay,
                        hipStream_t s) {
    for (int g0 = 0, grp = 0; g0 < n; g0 += OPT_MAX_TENSORS, ++grp) {
        AdamArgs a;
        a.n = (n - g0) < OPT_MAX_TENSORS ? (n - g0) : OPT_MAX_TENSORS;
        unsigned int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            a.p[i] = p[g0 + i]; a.g[i] = g[g0 + i]; a.m[i] = m[g0 + i]; a.v[i] = v[g0 + i]; a.step[i] = step[g0 + i];
            a.numel[i] = numel[g0 + i];
            a.first_block[i] = blocks;
            blocks += (unsigned int)((numel[g0 + i] + OPT_BLOCK_ELEMS - 1) / OPT_BLOCK_ELEMS);
        }
        a.first_block[a.n] = blocks;
        a.ticket = ticket + (size_t)grp * (OPT_SUBTICKETS + 1);
        a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
        a.log_beta1 = (float)log((double)beta1);
        a.log_beta2 = (float)log((double)beta2);
        if (blocks) adamw_multi_kernel<<<dim3(blocks), dim3(256), 0, s>>>(a);
    }
}

}  // namespace aecf
