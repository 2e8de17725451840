// Tail of the backward pass (gfx950): deterministic reduction of the float32 partial slabs and the
// rank-1 / matvec parameter gradients of the query and key projections.
//
//   dW_o, db_o, dW_v, db_v, u  <- sum over batch splits of the gemm_tn slabs        reduce_segments (1 launch)
//   dq'[j]   = scale * W_k[j,:] . u[h(j)]                                            fin_dqp
//   dW_q     = dq' (x) q ;  dW_k[j,:] = qs[j] * u[h(j)] ;  db_q = dq' ;  db_k = 0    fin_outer (+ dquery partials)
//   dquery   = W_q^T dq'                                                             fin_dquery
// db_k is exactly zero: dK = ds (x) qs and every softmax-backward row of ds sums to zero.
#include "aecf_kernels.h"

namespace aecf {

// 8 consecutive lanes per output element: lane g sums the splits k = g, g+8, ... (coalesced across elements is not
// needed: slabs are [split][element], so the 8 lanes read 8 different slabs at the same offset), then a fixed-order
// butterfly adds the 8 partials -> deterministic, and short loops even for hundreds of splits.
__global__ __launch_bounds__(256) void reduce_segments_kernel(ReduceSegs r) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int g = threadIdx.x & 7;
#pragma unroll
    for (int s = 0; s < ReduceSegs::N; ++s) {
        if (i < r.n[s]) {
            float a = 0.f;
            for (int k = g; k < r.splits[s]; k += 8) a += r.src[s][(int64_t)k * r.n[s] + i];
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            a += __shfl_xor(a, 4, 64);
            if (g == 0) r.dst[s][i] = a;
            return;
        }
        i -= r.n[s];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fin_dqp_kernel(FinalizeArgs p) {   // wave per j
    using X = Tr<T>;
    const int j = blockIdx.x * 4 + wave_id();
    if (j >= p.E) return;
    const int lane = lane_id();
    const typename X::elem* wk = reinterpret_cast<const typename X::elem*>(p.w_in) + (int64_t)p.E * p.E;
    const float* u = p.u + (int64_t)(j / p.hd) * p.E;
    float a = 0.f;
    for (int k = lane; k < p.E; k += 64) a += X::to_f32(wk[(int64_t)j * p.E + k]) * u[k];
    a = reduce_wave(a);
    if (lane == 0) p.dqp[j] = a * p.scale;
}

// grid (E/64 k-blocks, E/64 j-blocks); 256 threads = 64 k x 4 j-groups, 16 j per thread
template <typename T>
__global__ __launch_bounds__(256) void fin_outer_kernel(FinalizeArgs p) {
    using X = Tr<T>;
    __shared__ float red[4][64];
    const int E = p.E;
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int jg = threadIdx.x >> 6;
    const int jb = blockIdx.y;
    const typename X::elem* wq = reinterpret_cast<const typename X::elem*>(p.w_in);
    const float qk = X::to_f32(reinterpret_cast<const typename X::elem*>(p.query)[k]);
    float part = 0.f;
    for (int jj = jg; jj < 64; jj += 4) {
        const int j = jb * 64 + jj;
        const float dq = p.dqp[j];
        p.dw_in[(int64_t)j * E + k] = dq * qk;                                       // dW_q
        p.dw_in[(int64_t)(E + j) * E + k] = p.qs[j] * p.u[(int64_t)(j / p.hd) * E + k];   // dW_k
        part = fmaf(dq, X::to_f32(wq[(int64_t)j * E + k]), part);
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
            p.db_in[j] = dq;        // db_q
            p.db_in[E + j] = 0.f;   // db_k
        }
    }
    red[jg][threadIdx.x & 63] = part;
    __syncthreads();
    if (jg == 0) {
        const int t = threadIdx.x;
        p.dq_part[(int64_t)jb * E + k] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    }
}

__global__ __launch_bounds__(256) void fin_dquery_kernel(FinalizeArgs p) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= p.E) return;
    float a = 0.f;
    for (int jb = 0; jb < p.E / 64; ++jb) a += p.dq_part[(int64_t)jb * p.E + k];
    p.dquery[k] = a;
}

void launch_reduce_segments(const ReduceSegs& r, hipStream_t s) {
    int64_t total = 0;
    for (int i = 0; i < ReduceSegs::N; ++i) total += r.n[i];
    reduce_segments_kernel<<<dim3((unsigned)((total * 8 + 255) / 256)), dim3(256), 0, s>>>(r);
}

void launch_finalize(int dtype, const FinalizeArgs& a, hipStream_t s) {
    const int E = a.E;
    if (dtype == 0) {
        fin_dqp_kernel<BF16><<<dim3((E + 3) / 4), dim3(256), 0, s>>>(a);
        fin_outer_kernel<BF16><<<dim3(E / 64, E / 64), dim3(256), 0, s>>>(a);
    } else {
        fin_dqp_kernel<F32><<<dim3((E + 3) / 4), dim3(256), 0, s>>>(a);
        fin_outer_kernel<F32><<<dim3(E / 64, E / 64), dim3(256), 0, s>>>(a);
    }
    fin_dquery_kernel<<<dim3((E + 255) / 256), dim3(256), 0, s>>>(a);
}

}  // namespace aecf
