// Backward kernels of the AECF fusion pool (shared-query hot path), gfx950.
//
// With do = dy W_o and, per head h, g_h[b] = W_v,h^T do_h[b]  (an E-vector per (sample, head), never
// written to memory):
//   da[b,h,m] = g_h[b] . x[b,m]                 (+ do_h.b_v,h: constant over m, cancels below)
//   dp        = da + dwbar[b,m] / H
//   ds        = a * (dp - sum_m a dp)           softmax backward                    bwd_da  (MFMA + VALU)
//   dx[b,m]   = sum_h a[b,h,m] g_h[b] + sum_h ds[b,h,m] A[h]                        bwd_dx  (MFMA + VALU)
//   dW_o      = dy^T o,  db_o = sum_b dy                                            gemm_tn
//   dW_v,h    = do_h^T pooled_h,  db_v = sum_b do,  u[h] = sum_{b,m} ds[b,h,m] x[b,m]   gemm_tn (pooled)
//   dW_k,h    = qs_h (x) u[h];  db_k = 0;  dq' = scale * W_k,h u[h];  dW_q = dq' (x) q;  db_q = dq';
//   dquery    = W_q^T dq'                                                           finalize
// g_h is produced transposed (rows = E index, cols = samples) so that each lane of the accumulator
// holds 4 consecutive E positions of ONE sample: the products with x / the dx stores are then
// 8-byte (bf16) or 16-byte (f32) accesses.
#include "aecf_kernels.h"

namespace aecf {

// ------------------------------------------------------------------------------------------
// gemm_tn: out[j][k] = sum_b lhs[b][j] rhs[b][k] -- the reduction runs over the batch, which is the
// slow axis of both row-major operands.  Each lane loads an NR(batch rows) x 4(features) block with
// coalesced 8/16-byte loads and transposes it in registers into 4 MFMA fragments (one per feature):
// lane r16 of fragment f stands for feature 4*r16 + f, so an output tile holds the strided feature
// set {4*i + f}.  No LDS, no transposed reads.
template <typename T> struct Blk;
template <> struct Blk<BF16> {
    static constexpr int NR = 8;
    u32x2 r[8];
    __device__ __forceinline__ void zero_row(int t) { r[t] = u32x2{0u, 0u}; }
    __device__ __forceinline__ void load_row(int t, const unsigned short* p) { r[t] = *reinterpret_cast<const u32x2*>(p); }
    __device__ __forceinline__ void get_row(int t, float* v) const {
        v[0] = __uint_as_float(r[t][0] << 16);
        v[1] = __uint_as_float(r[t][0] & 0xffff0000u);
        v[2] = __uint_as_float(r[t][1] << 16);
        v[3] = __uint_as_float(r[t][1] & 0xffff0000u);
    }
    __device__ __forceinline__ void set_row(int t, const float* v) {
        r[t][0] = pack_bf16x2(v[0], v[1]);
        r[t][1] = pack_bf16x2(v[2], v[3]);
    }
    // fragment of feature f: element t = row t
    __device__ __forceinline__ u32x4 frag(int f) const {
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned int a = r[2 * d][f >> 1], b = r[2 * d + 1][f >> 1];
            o[d] = (f & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
        }
        return o;
    }
};
template <> struct Blk<F32> {
    static constexpr int NR = 4;
    f32x4 r[4];
    __device__ __forceinline__ void zero_row(int t) { r[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void load_row(int t, const float* p) { r[t] = *reinterpret_cast<const f32x4*>(p); }
    __device__ __forceinline__ void get_row(int t, float* v) const { v[0] = r[t][0]; v[1] = r[t][1]; v[2] = r[t][2]; v[3] = r[t][3]; }
    __device__ __forceinline__ void set_row(int t, const float* v) { r[t] = f32x4{v[0], v[1], v[2], v[3]}; }
    __device__ __forceinline__ f32x4 frag(int f) const { return f32x4{r[0][f], r[1][f], r[2][f], r[3][f]}; }
};

// grid (ceil(E/128) k-tiles, ceil(E/128) j-tiles, S); block 256 = 2(j) x 2(k) waves of 64 x 64
template <typename T, int M_, bool POOLED>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs p) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    typedef typename X::frag frag;
    constexpr int NR = Blk<T>::NR;            // batch rows per lane per step
    constexpr int STEP = 4 * NR;              // batch rows per wave step (32 bf16 / 16 f32)
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    const int E = p.E;
    const int jw0 = blockIdx.y * 128 + (w >> 1) * 64;
    const int kw0 = blockIdx.x * 128 + (w & 1) * 64;
    if (jw0 >= E || kw0 >= E) return;
    const int split = blockIdx.z;
    const int64_t rbeg = (int64_t)split * p.rows_per_split;
    const int64_t rend = (rbeg + p.rows_per_split) < p.B ? (rbeg + p.rows_per_split) : p.B;

    const int jf = jw0 + 4 * r16;             // this lane's 4 lhs features
    const int kf = kw0 + 4 * r16;             // this lane's 4 rhs features
    // heads touched by the 64 lhs features of this wave (pooled rhs differs per head)
    const int hA = POOLED ? jw0 / p.hd : 0;
    const int hB = POOLED ? (jw0 + 63 < E ? (jw0 + 63) : (E - 1)) / p.hd : 0;
    const int my_head = POOLED ? jf / p.hd : 0;
    const bool do_u = POOLED && blockIdx.y == 0 && (w >> 1) == 0;   // the j-tile-0 waves also reduce u

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 uacc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) uacc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};

    const elem* lhs = reinterpret_cast<const elem*>(p.lhs);
    const elem* rhs = reinterpret_cast<const elem*>(p.rhs);

    for (int64_t base = rbeg; base < rend; base += STEP) {
        Blk<T> L;
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const int64_t bb = base + NR * lg + t;
            if (bb < rend) L.load_row(t, lhs + bb * E + jf); else L.zero_row(t);
            float v[4];
            L.get_row(t, v);
            cs[0] += v[0]; cs[1] += v[1]; cs[2] += v[2]; cs[3] += v[3];
        }
        if (!POOLED) {
            Blk<T> R;
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const int64_t bb = base + NR * lg + t;
                if (bb < rend) R.load_row(t, rhs + bb * E + kf); else R.zero_row(t);
            }
            frag fb[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) fb[f] = R.frag(f);
#pragma unroll
            for (int fa = 0; fa < 4; ++fa) {
                frag af = L.frag(fa);
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[fa][f] = X::mma(af, fb[f], acc[fa][f]);
            }
        } else {
            // raw x blocks of every modality (kept for u) and the per-head pooled block
            Blk<T> XB[M_];
#pragma unroll
            for (int m = 0; m < M_; ++m)
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    const int64_t bb = base + NR * lg + t;
                    if (bb < rend) XB[m].load_row(t, rhs + (bb * M_ + m) * E + kf); else XB[m].zero_row(t);
                }
            for (int h = hA; h <= hB; ++h) {
                Blk<T> R;
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    const int64_t bb = base + NR * lg + t;
                    float pv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (bb < rend) {
#pragma unroll
                        for (int m = 0; m < M_; ++m) {
                            const float pm = p.probs[(bb * p.H + h) * M_ + m];
                            float xv[4];
                            XB[m].get_row(t, xv);
                            pv[0] = fmaf(pm, xv[0], pv[0]); pv[1] = fmaf(pm, xv[1], pv[1]);
                            pv[2] = fmaf(pm, xv[2], pv[2]); pv[3] = fmaf(pm, xv[3], pv[3]);
                        }
                    }
                    R.set_row(t, pv);
                }
                frag fb[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) fb[f] = R.frag(f);
                // lhs rows that belong to another head contribute nothing for this rhs
                Blk<T> Lh = L;
                if (my_head != h) {
#pragma unroll
                    for (int t = 0; t < NR; ++t) Lh.zero_row(t);
                }
#pragma unroll
                for (int fa = 0; fa < 4; ++fa) {
                    frag af = Lh.frag(fa);
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[fa][f] = X::mma(af, fb[f], acc[fa][f]);
                }
            }
            if (do_u) {
                // u[h][k] += sum_{rows, m} ds[row][h][m] x[row][m][k]: A operand = ds (row index r16 = head)
#pragma unroll
                for (int m = 0; m < M_; ++m) {
                    float dv[X::EPL], dl[X::EPL];
#pragma unroll
                    for (int t = 0; t < NR; ++t) {
                        const int64_t bb = base + NR * lg + t;
                        dv[t] = (bb < rend && r16 < p.H) ? p.dsbuf[(bb * p.H + r16) * M_ + m] : 0.f;
                    }
                    frag dhi = X::pack(dv);
                    frag dlo = dhi;
                    if (X::BYTES == 2) {   // bf16 MFMA inputs: keep ds to ~16 bits with a hi/lo pair
                        float hv[X::EPL];
                        X::unpack(dhi, hv);
#pragma unroll
                        for (int t = 0; t < NR; ++t) dl[t] = dv[t] - hv[t];
                        dlo = X::pack(dl);
                    }
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        frag xb = XB[m].frag(f);
                        uacc[f] = X::mma(dhi, xb, uacc[f]);
                        if (X::BYTES == 2) uacc[f] = X::mma(dlo, xb, uacc[f]);
                    }
                }
            }
        }
    }

    // slab stores: tile (fa, fb): rows i = 4*lg + r -> j = jw0 + 4*i + fa ; col r16 -> k = kw0 + 4*r16 + fb
    float* out = p.out + (int64_t)split * E * E;
#pragma unroll
    for (int fa = 0; fa < 4; ++fa)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jw0 + 4 * (4 * lg + r) + fa;
                const int k = kw0 + 4 * r16 + f;
                out[(int64_t)j * E + k] = acc[fa][f][r];
            }
    if (blockIdx.x == 0 && (w & 1) == 0 && p.colsum) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float v = reduce_lg(cs[f]);
            if (lg == 0) p.colsum[(int64_t)split * E + jf + f] = v;
        }
    }
    if (do_u) {
        float* u = p.u + (int64_t)split * HPAD * E;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) u[(int64_t)(4 * lg + r) * E + kw0 + 4 * r16 + f] = uacc[f][r];
    }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n,
                                                           int S) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += src[(int64_t)s * n + i];
    dst[i] = a;
}

// ------------------------------------------------------------------------------------------
// finalize: the rank-1 / matvec tail of the parameter gradients
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

template <typename T>
__global__ __launch_bounds__(256) void fin_outer_kernel(FinalizeArgs p) {   // grid (E/256, E): row j, 256 k per block
    using X = Tr<T>;
    const int j = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= p.E) return;
    const int E = p.E;
    const float qk = X::to_f32(reinterpret_cast<const typename X::elem*>(p.query)[k]);
    p.dw_in[(int64_t)j * E + k] = p.dqp[j] * qk;                                        // dW_q
    p.dw_in[(int64_t)(E + j) * E + k] = p.qs[j] * p.u[(int64_t)(j / p.hd) * E + k];     // dW_k
    if (k == 0) {
        p.db_in[j] = p.dqp[j];     // db_q
        p.db_in[E + j] = 0.f;      // db_k = qs_h * sum_m ds = 0 exactly (softmax rows sum to 1)
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fin_dquery_kernel(FinalizeArgs p) {  // grid E/64; 4 j-groups x 64 k
    using X = Tr<T>;
    __shared__ float red[4][64];
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int jg = threadIdx.x >> 6;
    const typename X::elem* wq = reinterpret_cast<const typename X::elem*>(p.w_in);
    float a = 0.f;
    for (int j = jg; j < p.E; j += 4) a += p.dqp[j] * X::to_f32(wq[(int64_t)j * p.E + k]);
    red[jg][threadIdx.x & 63] = a;
    __syncthreads();
    if (jg == 0) {
        const int t = threadIdx.x;
        p.dquery[k] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    }
}

// ------------------------------------------------------------------------------------------
void launch_gemm_tn(int dtype, const GemmTnArgs& a, hipStream_t s) {
    dim3 grid((a.E + 127) / 128, (a.E + 127) / 128, a.splits), block(256);
    if (!a.pooled) {
        if (dtype == 0)
            gemm_tn_kernel<BF16, 1, false><<<grid, block, 0, s>>>(a);
        else
            gemm_tn_kernel<F32, 1, false><<<grid, block, 0, s>>>(a);
        return;
    }
    AECF_DISPATCH_M(a.M, {
        if (dtype == 0)
            gemm_tn_kernel<BF16, M_, true><<<grid, block, 0, s>>>(a);
        else
            gemm_tn_kernel<F32, M_, true><<<grid, block, 0, s>>>(a);
    });
}

void launch_reduce_slabs(const float* src, float* dst, int64_t n, int S, hipStream_t s) {
    reduce_slabs_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(src, dst, n, S);
}

void launch_finalize(int dtype, const FinalizeArgs& a, hipStream_t s) {
    const int E = a.E;
    if (dtype == 0) {
        fin_dqp_kernel<BF16><<<dim3((E + 3) / 4), dim3(256), 0, s>>>(a);
        fin_outer_kernel<BF16><<<dim3((E + 255) / 256, E), dim3(256), 0, s>>>(a);
        fin_dquery_kernel<BF16><<<dim3(E / 64), dim3(256), 0, s>>>(a);
    } else {
        fin_dqp_kernel<F32><<<dim3((E + 3) / 4), dim3(256), 0, s>>>(a);
        fin_outer_kernel<F32><<<dim3((E + 255) / 256, E), dim3(256), 0, s>>>(a);
        fin_dquery_kernel<F32><<<dim3(E / 64), dim3(256), 0, s>>>(a);
    }
}

}  // namespace aecf
