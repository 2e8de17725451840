// Tail of the backward pass (gfx950): deterministic reduction of the float32 partial slabs and the
// rank-1 / matvec parameter gradients of the query and key projections.
//
//   dW_o, db_o, dW_v, db_v, u  <- sum over batch splits of the gemm_tn slabs        reduce_segments (1 launch)
//   dq'[j]   = scale * W_k[j,:] . u[h(j)]                                            fin_outer (per block)
//   dW_q     = dq' (x) q ;  dW_k[j,:] = qs[j] * u[h(j)] ;  db_q = dq' ;  db_k = 0    fin_outer (+ dquery partials)
//   dquery   = W_q^T dq'                                                             fin_dquery
// db_k is exactly zero: dK = ds (x) qs and every softmax-backward row of ds sums to zero.
#include <stdlib.h>
#include "aecf_kernels.h"

namespace aecf {

// One wave reduces 64 consecutive output elements: lane = (g = lane >> 4, e = lane & 15) reads the float4 at elements
// 4e..4e+3 of the splits k = g, g+4, g+8, ... (a 16-lane group reads 256 contiguous bytes of one slab), then a
// fixed-order butterfly over the 4 groups adds the partials -> deterministic, coalesced, short loops even for
// hundreds of splits.  Every segment length is a multiple of 4.
__global__ __launch_bounds__(256) void reduce_segments_kernel(ReduceSegs r) {
    const int lane = threadIdx.x & 63, g = lane >> 4, e = lane & 15;
    int64_t q = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + e;      // float4 index over all segments
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

// dq'[j] = scale * W_k[j,:] . u[h(j)]  for the 16 rows j of this block (wave per 4 rows), then
// dW_q[j][k] = dq'[j] q[k], dW_k[j][k] = qs[j] u[h(j)][k], db_q = dq', db_k = 0 and the dquery partial
// sum_j dq'[j] W_q[j][k] of these 16 rows.  grid (E/64 k-blocks, E/16 j-blocks); 256 threads = 64 k x 4 j-groups.
// kc: 64-column slices per block (the dq' dot of a row block is repeated by every block of its row: wide embeddings
// take several slices per block so that the repetition stays at E / (64 kc))
template <typename T>
__global__ __launch_bounds__(256) void fin_outer_kernel(FinalizeArgs p, int kc) {
    using X = Tr<T>;
    __shared__ float red[4][64];
    __shared__ float dql[16];
    const int E = p.E;
    const int jg = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int jb = blockIdx.y;
    const typename X::elem* wq = reinterpret_cast<const typename X::elem*>(p.w_in);
    const typename X::elem* wk = wq + (int64_t)E * E;
    {   // 16 threads per row j: contiguous E/16-element pieces (16-byte loads, all independent), 4-step butterfly
        const int jj = threadIdx.x >> 4, part = threadIdx.x & 15;
        const int j = jb * 16 + jj;
        const int plen = E / 16;                                   // multiple of 4 (E % 64 == 0)
        const float* u = p.u + (int64_t)(j / p.hd) * E + part * plen;
        const typename X::elem* wr = wk + (int64_t)j * E + part * plen;
        float a = 0.f;
        for (int kk = 0; kk < plen; kk += 4) {
            float wv[4];
            X::load4(wr + kk, wv);
            const f32x4 uv = *reinterpret_cast<const f32x4*>(u + kk);
#pragma unroll
            for (int e = 0; e < 4; ++e) a = fmaf(wv[e], uv[e], a);
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        a += __shfl_xor(a, 8, 64);
        if (part == 0) dql[jj] = a * p.scale;
    }
    __syncthreads();
    for (int c = 0; c < kc; ++c) {
        const int k = (blockIdx.x * kc + c) * 64 + lane;
        const float qk = X::to_f32(reinterpret_cast<const typename X::elem*>(p.query)[k]);
        float part = 0.f;
        for (int jj = jg; jj < 16; jj += 4) {
            const int j = jb * 16 + jj;
            const float dq = dql[jj];
            store_grad(p.dw_in, (int64_t)j * E + k, dq * qk, p.grad_bf16);                                         // dW_q
            store_grad(p.dw_in, (int64_t)(E + j) * E + k, p.qs[j] * p.u[(int64_t)(j / p.hd) * E + k], p.grad_bf16);   // dW_k
            part = fmaf(dq, X::to_f32(wq[(int64_t)j * E + k]), part);
            if (blockIdx.x == 0 && c == 0 && lane == 0) {
                store_grad(p.db_in, j, dq, p.grad_bf16);         // db_q
                store_grad(p.db_in, E + j, 0.f, p.grad_bf16);    // db_k
            }
        }
        if (c > 0) __syncthreads();
        red[jg][lane] = part;
        __syncthreads();
        if (jg == 0) p.dq_part[(int64_t)jb * E + k] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    }
}

__global__ __launch_bounds__(64) void fin_dquery_kernel(FinalizeArgs p) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= p.E) return;
    float a = 0.f;
#pragma unroll 8
    for (int jb = 0; jb < p.E / 16; ++jb) a += p.dq_part[(int64_t)jb * p.E + k];
    store_grad(p.dquery, k, a, p.grad_bf16);
}

void launch_reduce_segments(const ReduceSegs& r, hipStream_t s) {
    int64_t waves = 0;                                   // one wave per 64 elements, segments padded to waves
    for (int i = 0; i < ReduceSegs::N; ++i) waves += (r.n[i] + 63) / 64;
    reduce_segments_kernel<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s>>>(r);
}

void launch_finalize(int dtype, const FinalizeArgs& a, hipStream_t s) {
    const int E = a.E;
    int kc = E >= 1024 ? 4 : 1;                                   // keep >= 256 blocks
    while ((E / 64) % kc != 0) --kc;
    if (dtype == 0) {
        fin_outer_kernel<BF16><<<dim3(E / 64 / kc, E / 16), dim3(256), 0, s>>>(a, kc);
    } else {
        fin_outer_kernel<F32><<<dim3(E / 64 / kc, E / 16), dim3(256), 0, s>>>(a, kc);
    }
    fin_dquery_kernel<<<dim3((E + 63) / 64), dim3(64), 0, s>>>(a);
}

}  // namespace aecf
