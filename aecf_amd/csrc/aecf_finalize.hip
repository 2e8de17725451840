// Tail of the backward pass (gfx950): deterministic reduction of the float32 partial slabs and the
// rank-1 / matvec parameter gradients of the query and key projections.
//
//   dW_o, db_o, dW_v, db_v     <- sum over batch splits of the gemm_tn slabs         R blocks
//   dq'[j]   = scale * W_k[j,:] . u[h(j)]                                            Q blocks (their 16 rows), D blocks (all)
//   dW_q     = dq' (x) q ;  dW_k[j,:] = qs[j] * u[h(j)] ;  db_q = dq' ;  db_k = 0    Q blocks
//   dquery   = W_q^T dq'                                                             D blocks
// db_k is exactly zero: dK = ds (x) qs and every softmax-backward row of ds sums to zero.
// ONE launch (round 4; three before: reduce_segments, fin_outer, fin_dquery = 27 us at the headline shape).  What made them
// three was a chain of cross-block dependencies: slabs -> u -> dq' -> dquery.  It is cut without any in-kernel hand-off (a
// device-scope release on this multi-XCD part writes back an XCD's L2: measured 15 us, profiles/r01_pmc_notes.md):
//   * u [H, E] arrives already reduced -- 16 KB that every block can read: the dx kernel, which runs between the score
//     gradient that writes the u slabs and this launch, adds them up as a side job of its weight prologue (dx_ws2_kernel;
//     other shapes: one small reduce launch);
//   * dq' [E] arrives computed as well: E dots of length E against that u, a side job of the dW_v launch that runs between
//     the dx kernel and this one (DqpJob, aecf_common.h; other shapes: launch_dqp).  (First form of this launch: the E / 64
//     dquery blocks each formed all E dots themselves -- 74 us for the launch, a serial chain of memory round trips.)
// So the tail is: score gradient (u slabs) -> dx (+ u reduced) -> dW_v (+ dq') -> this launch, nothing in it waits for
// another block.
#include <stdlib.h>
#include "aecf_kernels.h"

namespace aecf {

// One wave reduces 64 consecutive output elements: lane = (g = lane >> 4, e = lane & 15) reads the float4 at elements
// 4e..4e+3 of the splits k = g, g+4, g+8, ... (a 16-lane group reads 256 contiguous bytes of one slab), then a
// fixed-order butterfly over the 4 groups adds the partials -> deterministic, coalesced, short loops even for
// hundreds of splits.  Every segment length is a multiple of 4.
__device__ __forceinline__ void reduce_segments_block(const ReduceSegs& r, int64_t block) {
    const int lane = threadIdx.x & 63, g = lane >> 4, e = lane & 15;
    int64_t q = (block * 4 + (threadIdx.x >> 6)) * 16 + e;                     // float4 index over all segments
#pragma unroll
    for (int s = 0; s < ReduceSegs::N; ++s) {
        const int64_t nq = (r.n[s] + 63) / 64 * 16;                            // segments start on a wave boundary
        if (q < nq) {
            const bool on = 4 * q < r.n[s];
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on) {
                const float* base = r.src[s] + 4 * q;
                const int64_t n = r.n[s];
                int k = g;
                for (; k + 28 < r.splits[s]; k += 32) {           // 8 independent loads in flight, summed in order
                    f32x4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(base + (int64_t)(k + 4 * j) * n);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += v[j];
                }
                for (; k < r.splits[s]; k += 4) a += *reinterpret_cast<const f32x4*>(base + (int64_t)k * n);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[c] += __shfl_xor(a[c], 16, 64);
                a[c] += __shfl_xor(a[c], 32, 64);
            }
            if (on && g == 0) {
                a *= r.scale[s];
                if (r.dst_bf16[s]) {
                    u32x2 o;
                    o[0] = pack_bf16x2(a[0], a[1]);
                    o[1] = pack_bf16x2(a[2], a[3]);
                    *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(r.dst[s]) + 4 * q) = o;
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(r.dst[s]) + 4 * q) = a;
                }
            }
            return;
        }
        q -= nq;
    }
}

__global__ __launch_bounds__(256) void reduce_segments_kernel(ReduceSegs r) { reduce_segments_block(r, blockIdx.x); }

// dW_q[j][k] = dq'[j] q[k], dW_k[j][k] = qs[j] u[h(j)][k], db_q = dq', db_k = 0 for the 16 rows j x 64 kc columns of this block;
// 256 threads = 64 k x 4 j-groups.  dq' arrives computed (DqpJob).
template <typename T>
__device__ __forceinline__ void fin_outer_block(const FinalizeArgs& p, int kc, int bx, int jb) {
    using X = Tr<T>;
    const int E = p.E;
    const int jg = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = 0; c < kc; ++c) {
        const int k = (bx * kc + c) * 64 + lane;
        const float qk = X::to_f32(reinterpret_cast<const typename X::elem*>(p.query)[k]);
        for (int jj = jg; jj < 16; jj += 4) {
            const int j = jb * 16 + jj;
            const float dq = p.dqp[j] * p.gscale;
            store_grad(p.dw_in, (int64_t)j * E + k, dq * qk, p.grad_bf16);                                         // dW_q
            store_grad(p.dw_in, (int64_t)(E + j) * E + k, p.qs[j] * p.gscale * p.u[(int64_t)(j / p.hd) * E + k], p.grad_bf16);   // dW_k
            if (bx == 0 && c == 0 && lane == 0) {
                store_grad(p.db_in, j, dq, p.grad_bf16);         // db_q
                store_grad(p.db_in, E + j, 0.f, p.grad_bf16);    // db_k
            }
        }
    }
}

// dquery[k] = sum_j dq'[j] W_q[j][k] for the 64 columns k of this block: thread = (row lane t >> 3, 8-column piece t & 7), rows
// t >> 3 + 32 i all in flight at once (16 per pass: one memory round trip per 512 rows), then the 32 row lanes are folded in
// fixed order through LDS
template <typename T>
__device__ __forceinline__ void fin_dquery_block(const FinalizeArgs& p, int kb, float* scratch) {
    using X = Tr<T>;
    const int E = p.E;
    const int sub = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const typename X::elem* wq = reinterpret_cast<const typename X::elem*>(p.w_in) + kb * 64 + 8 * sub;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const int npass = E / 32;                                     // rows per row lane (E % 64 == 0: even, any count)
    for (int i0 = 0; i0 < npass; i0 += 16) {
        float wv[16][8], dq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool on = i0 + i < npass;                       // (branch-free: an always-valid row, weight 0 beyond the end)
            const int j = rl + 32 * (on ? i0 + i : 0);
            X::load4(wq + (int64_t)j * E, wv[i]);
            X::load4(wq + (int64_t)j * E + 4, wv[i] + 4);
            dq[i] = on ? p.dqp[j] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(dq[i], wv[i][e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) scratch[rl * 64 + 8 * sub + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) t += scratch[r * 64 + threadIdx.x];
        store_grad(p.dquery, kb * 64 + threadIdx.x, t * p.gscale, p.grad_bf16);
    }
}

// block roles by id: [0, nd) dquery column blocks, [nd, nd + nqx nqy) outer-product blocks, the rest slab reductions
template <typename T>
__global__ __launch_bounds__(256) void finalize_all_kernel(FinalizeArgs p, ReduceSegs r, int kc, int nd, int nqx, int nqy) {
    __shared__ float scratch[32 * 64];
    const int b = blockIdx.x;
    if (b < nd) { fin_dquery_block<T>(p, b, scratch); return; }
    const int q = b - nd;
    if (q < nqx * nqy) { fin_outer_block<T>(p, kc, q % nqx, q / nqx); return; }
    reduce_segments_block(r, (int64_t)(q - nqx * nqy));
}

template <typename T>
__global__ __launch_bounds__(256) void dqp_kernel(DqpJob q) { dqp_rows<T>(q, blockIdx.x, gridDim.x); }

void launch_dqp(int dtype, const DqpJob& q, hipStream_t s) {
    const unsigned grid = (unsigned)((q.E + 3) / 4);               // a wave per row
    if (dtype == 0) dqp_kernel<BF16><<<dim3(grid), dim3(256), 0, s>>>(q);
    else dqp_kernel<F32><<<dim3(grid), dim3(256), 0, s>>>(q);
}

void launch_reduce_segments(const ReduceSegs& r, hipStream_t s) {
    int64_t waves = 0;                                   // one wave per 64 elements, segments padded to waves
    for (int i = 0; i < ReduceSegs::N; ++i) waves += (r.n[i] + 63) / 64;
    if (waves == 0) return;
    reduce_segments_kernel<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s>>>(r);
}

// the whole tail of the backward in one launch; a.u = the REDUCED u [H, E]; the segments of r are reduced as by
// launch_reduce_segments (pass n = 0 for the u segment: it is an input here)
void launch_finalize_all(int dtype, const FinalizeArgs& a, const ReduceSegs& r, hipStream_t s) {
    const int E = a.E;
    int kc = E >= 1024 ? 4 : 1;                                   // (the dq' dot of a row block is repeated E / (64 kc) times)
    while ((E / 64) % kc != 0) --kc;
    int64_t waves = 0;
    for (int i = 0; i < ReduceSegs::N; ++i) waves += (r.n[i] + 63) / 64;
    const int nd = E / 64, nqx = E / 64 / kc, nqy = E / 16;
    const unsigned grid = (unsigned)(nd + nqx * nqy + (waves + 3) / 4);
    if (dtype == 0) finalize_all_kernel<BF16><<<dim3(grid), dim3(256), 0, s>>>(a, r, kc, nd, nqx, nqy);
    else finalize_all_kernel<F32><<<dim3(grid), dim3(256), 0, s>>>(a, r, kc, nd, nqx, nqy);
}

}  // namespace aecf
