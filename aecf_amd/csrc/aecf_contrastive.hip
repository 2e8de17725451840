// Contrastive (InfoNCE) term of the AECF training objective (gfx950).  NOT in the reference:
// build-defined symmetric InfoNCE on L2-normalised fused embeddings with cross-batch (all-gathered) negatives.
//
// One direction, for the local query rows q [b,d] against all keys k [B_all,d] (both unit-norm):
//     S = q k^T (MFMA GEMM, float32 out) ; logits = S / T ; loss_i = logsumexp_j(logits_ij) - logits_{i, off+i}
//     G = (softmax(logits) - onehot) * coef / T              (coef = upstream scale, e.g. 0.5 / B_all)
//     dq = G k            (gemm_nt against k^T)              dk = G^T q   (batch-reduction gemm_tn, rectangular)
// The row kernel below does the softmax / loss / G in three sweeps of a logits row; the GEMMs are the library's
// MFMA kernels.  The symmetric term is the same call with the roles of the two views swapped.
#include "aecf_kernels.h"

namespace aecf {

// z -> z / max(||z||, eps), one wave per row
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(int64_t n, int d, float eps, const typename Tr<T>::elem* __restrict__ z,
                                                         typename Tr<T>::elem* __restrict__ zn, float* __restrict__ inv_norm) {
    using X = Tr<T>;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave_id();
    if (row >= n) return;
    const int lane = lane_id();
    float ss = 0.f;
    for (int e = lane; e < d; e += 64) { const float v = X::to_f32(z[row * d + e]); ss = fmaf(v, v, ss); }
    ss = reduce_wave(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
    for (int e = lane; e < d; e += 64) zn[row * d + e] = X::from_f32(X::to_f32(z[row * d + e]) * inv);
    if (lane == 0) inv_norm[row] = inv;
}

// dz = (dzn - zn (dzn . zn)) * inv_norm
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(int64_t n, int d, const typename Tr<T>::elem* __restrict__ zn,
                                                         const float* __restrict__ inv_norm, const float* __restrict__ dzn,
                                                         typename Tr<T>::elem* __restrict__ dz) {
    using X = Tr<T>;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave_id();
    if (row >= n) return;
    const int lane = lane_id();
    float dot = 0.f;
    for (int e = lane; e < d; e += 64) dot = fmaf(dzn[row * d + e], X::to_f32(zn[row * d + e]), dot);
    dot = reduce_wave(dot);
    const float inv = inv_norm[row];
    for (int e = lane; e < d; e += 64)
        dz[row * d + e] = X::from_f32((dzn[row * d + e] - X::to_f32(zn[row * d + e]) * dot) * inv);
}

__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
    v = fmaxf(v, __shfl_xor(v, 8, 64)); v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64));
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
    v = reduce_wave(v);
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one block per local row
template <typename T>
__global__ __launch_bounds__(256) void nce_rows_kernel(int64_t cols, int64_t row_offset, float inv_temp, float coef,
                                                       const float* __restrict__ S, typename Tr<T>::elem* __restrict__ G,
                                                       float* __restrict__ loss_rows) {
    using X = Tr<T>;
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float* s = S + row * cols;
    typename X::elem* g = G + row * cols;
    float mx = -INFINITY;
    for (int64_t j = threadIdx.x; j < cols; j += 256) mx = fmaxf(mx, s[j] * inv_temp);
    mx = block_reduce_max(mx, red);
    float sum = 0.f;
    for (int64_t j = threadIdx.x; j < cols; j += 256) sum += expf(s[j] * inv_temp - mx);
    sum = block_reduce_sum(sum, red);
    const float lse = mx + logf(sum);
    const int64_t pos = row_offset + row;
    const float scale = coef * inv_temp;
    for (int64_t j = threadIdx.x; j < cols; j += 256) {
        const float pj = expf(s[j] * inv_temp - lse);
        g[j] = X::from_f32((pj - (j == pos ? 1.0f : 0.0f)) * scale);
    }
    if (threadIdx.x == 0) loss_rows[row] = lse - s[pos] * inv_temp;
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_rect_kernel(const typename Tr<T>::elem* __restrict__ src,
                                                             typename Tr<T>::elem* __restrict__ dst, int64_t R, int64_t C) {
    __shared__ typename Tr<T>::elem tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = src[(r0 + r) * C + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) dst[(c0 + r) * R + r0 + tx] = tile[tx][r];
}

void launch_l2norm_fwd(int dtype, int64_t n, int d, float eps, const void* z, void* zn, float* inv_norm, hipStream_t s) {
    dim3 grid((unsigned)((n + 3) / 4)), block(256);
    if (dtype == 0) l2norm_fwd_kernel<BF16><<<grid, block, 0, s>>>(n, d, eps, (const unsigned short*)z, (unsigned short*)zn, inv_norm);
    else l2norm_fwd_kernel<F32><<<grid, block, 0, s>>>(n, d, eps, (const float*)z, (float*)zn, inv_norm);
}

void launch_l2norm_bwd(int dtype, int64_t n, int d, const void* zn, const float* inv_norm, const float* dzn, void* dz,
                       hipStream_t s) {
    dim3 grid((unsigned)((n + 3) / 4)), block(256);
    if (dtype == 0) l2norm_bwd_kernel<BF16><<<grid, block, 0, s>>>(n, d, (const unsigned short*)zn, inv_norm, dzn, (unsigned short*)dz);
    else l2norm_bwd_kernel<F32><<<grid, block, 0, s>>>(n, d, (const float*)zn, inv_norm, dzn, (float*)dz);
}

void launch_nce_rows(int dtype, int64_t rows, int64_t cols, int64_t row_offset, float inv_temp, float coef, const float* S,
                     void* G, float* loss_rows, hipStream_t s) {
    dim3 grid((unsigned)rows), block(256);
    if (dtype == 0) nce_rows_kernel<BF16><<<grid, block, 0, s>>>(cols, row_offset, inv_temp, coef, S, (unsigned short*)G, loss_rows);
    else nce_rows_kernel<F32><<<grid, block, 0, s>>>(cols, row_offset, inv_temp, coef, S, (float*)G, loss_rows);
}

void launch_transpose_rect(int dtype, const void* src, void* dst, int64_t R, int64_t C, hipStream_t s) {
    dim3 grid((unsigned)(C / 32), (unsigned)(R / 32)), block(256);
    if (dtype == 0) transpose_rect_kernel<BF16><<<grid, block, 0, s>>>((const unsigned short*)src, (unsigned short*)dst, R, C);
    else transpose_rect_kernel<F32><<<grid, block, 0, s>>>((const float*)src, (float*)dst, R, C);
}

}  // namespace aecf
