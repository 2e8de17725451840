// Forward kernels of the AECF fusion pool (shared-query hot path), gfx950.
//
// Algebra (exact re-association of torch multi_head_attention_forward, functional.py:6576-6612,
// for ONE query shared by the whole batch -- ref aecf/AECFLayer.py:694-695):
//   qs      = (W_q q + b_q) / sqrt(hd)                                   [E]        prep_qs
//   A[h]    = W_k,h^T qs_h                                               [H,E]      prep_amat
//   s[b,h,m]= x[b,m,:] . A[h]   (+ b_k,h.qs_h, constant over m: cancels in the softmax)
//   a       = softmax_m(s);  wbar[b,m] = mean_h a[b,h,m]                            gate_fwd (MFMA)
//   o[b,hj] = W_v,h (sum_m a[b,h,m] x[b,m,:]) + b_v                                 gemm_nt (pooled A operand, MFMA)
//   y       = o W_o^T + b_o                                                         gemm_nt
// The K projection never materialises; the V projection runs once per sample instead of once per
// (sample, modality).
#include <type_traits>
#include "aecf_kernels.h"

namespace aecf {

// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void prep_qs_kernel(const typename Tr<T>::elem* __restrict__ w_in,
                                                      const typename Tr<T>::elem* __restrict__ b_in,
                                                      const typename Tr<T>::elem* __restrict__ q, float* __restrict__ qs,
                                                      int E, float scale) {
    using X = Tr<T>;
    const int j = blockIdx.x * 4 + wave_id();
    if (j >= E) return;
    const int lane = lane_id();
    float acc = 0.f;
    for (int k = lane; k < E; k += 64) acc += X::to_f32(w_in[(int64_t)j * E + k]) * X::to_f32(q[k]);
    acc = reduce_wave(acc);
    if (lane == 0) qs[j] = (acc + (b_in ? X::to_f32(b_in[j]) : 0.f)) * scale;
}

// grid (E/64, HPAD); block 256 = 4 j-groups x 64 k
template <typename T>
__global__ __launch_bounds__(256) void prep_amat_kernel(const typename Tr<T>::elem* __restrict__ w_in,
                                                        const float* __restrict__ qs, float* __restrict__ a_f32,
                                                        typename Tr<T>::elem* __restrict__ a_hi,
                                                        typename Tr<T>::elem* __restrict__ a_lo, int E, int H) {
    using X = Tr<T>;
    __shared__ float red[4][64];
    const int h = blockIdx.y;
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int jg = threadIdx.x >> 6;
    const int hd = E / H;
    float acc = 0.f;
    if (h < H) {
        const typename X::elem* wk = w_in + (int64_t)E * E;   // W_k block of the packed in_proj_weight
        for (int j = jg; j < hd; j += 4) acc += qs[h * hd + j] * X::to_f32(wk[(int64_t)(h * hd + j) * E + k]);
    }
    red[jg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (jg == 0) {
        const int t = threadIdx.x;
        float v = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        a_f32[h * E + k] = v;
        typename X::elem hi = X::from_f32(v);
        a_hi[h * E + k] = hi;
        if (a_lo) a_lo[h * E + k] = X::from_f32(v - X::to_f32(hi));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const typename Tr<T>::elem* __restrict__ src,
                                                        typename Tr<T>::elem* __restrict__ dst, int E) {
    __shared__ typename Tr<T>::elem tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int j0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8)
        if (j0 + r < E && k0 + tx < E) tile[r][tx] = src[(int64_t)(j0 + r) * E + k0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < E && j0 + tx < E) dst[(int64_t)(k0 + r) * E + j0 + tx] = tile[tx][r];
}

// All parameter-only preparation in ONE launch (each tiny kernel costs ~5 us of launch/drain on this part):
//   blocks [0, nA): the folded key matrix A[h][k] = sum_{j in head h} qs[j] W_k[j][k]; each block first forms the
//                   hd values qs[j] = (W_q[j,:] . q + b_q[j]) * scale of its head in LDS (same arithmetic order as
//                   prep_qs_kernel), the k-block 0 of each head also writes them out;
//   blocks [nA, ..): 32x32 tile transposes of up to two E x E matrices (backward: W_v^T, W_o^T).
template <typename T>
__global__ __launch_bounds__(256) void prep_all_kernel(const typename Tr<T>::elem* __restrict__ w_in,
                                                       const typename Tr<T>::elem* __restrict__ b_in,
                                                       const typename Tr<T>::elem* __restrict__ q, float scale,
                                                       float* __restrict__ qs, float* __restrict__ a_f32,
                                                       typename Tr<T>::elem* __restrict__ a_hi,
                                                       typename Tr<T>::elem* __restrict__ a_lo,
                                                       const typename Tr<T>::elem* __restrict__ t_src0,
                                                       typename Tr<T>::elem* __restrict__ t_dst0,
                                                       const typename Tr<T>::elem* __restrict__ t_src1,
                                                       typename Tr<T>::elem* __restrict__ t_dst1, int E, int H, FragJobs fj) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    __shared__ float red[4][64];
    __shared__ float qsl[1024];
    const int nA = (E / 64) * HPAD;
    int id = blockIdx.x;
    if (id < nA) {
        const int h = id / (E / 64), kb = id % (E / 64);
        const int k = kb * 64 + (threadIdx.x & 63);
        const int jg = threadIdx.x >> 6;
        const int hd = E / H;
        float acc = 0.f;
        if (h < H) {
            // 4 lanes per row j: each sums a contiguous quarter of the row with 16-byte loads (all independent, so
            // the whole head is one round of memory latency), then a 2-step butterfly
            const int part = threadIdx.x & 3;
            const int qlen = E / 4;                              // multiple of 16 elements
            for (int jj = threadIdx.x >> 2; jj < hd; jj += 64) {
                const int j = h * hd + jj;
                const elem* wr = w_in + (int64_t)j * E + part * qlen;
                const elem* qr = q + part * qlen;
                // (pieces in batches of 8 / 2, all of a batch's loads in flight before the first use: as a plain loop hipcc
                //  waited for every piece in turn -- E / 32 dependent memory round trips, most of this launch's 26 us at E = 1024;
                //  the arithmetic order is unchanged: one fmaf chain over kk)
                float a = 0.f;
                int kk = 0;
                auto batch = [&](auto nb) {
                    constexpr int NB = decltype(nb)::value;
                    typename X::frag wf[NB], qf[NB];
#pragma unroll
                    for (int i = 0; i < NB; ++i) { wf[i] = X::load(wr + kk + X::EPL * i); qf[i] = X::load(qr + kk + X::EPL * i); }
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        float wv[X::EPL], qv[X::EPL];
                        X::unpack(wf[i], wv);
                        X::unpack(qf[i], qv);
#pragma unroll
                        for (int e = 0; e < X::EPL; ++e) a = fmaf(wv[e], qv[e], a);
                    }
                    kk += X::EPL * NB;
                };
                while (kk + 8 * X::EPL <= qlen) batch(std::integral_constant<int, 8>{});
                while (kk + 2 * X::EPL <= qlen) batch(std::integral_constant<int, 2>{});
                while (kk < qlen) batch(std::integral_constant<int, 1>{});
                a += __shfl_xor(a, 1, 64);
                a += __shfl_xor(a, 2, 64);
                if (part == 0) {
                    const float v = (a + (b_in ? X::to_f32(b_in[j]) : 0.f)) * scale;
                    qsl[jj] = v;
                    if (kb == 0) qs[j] = v;
                }
            }
            __syncthreads();
            const elem* wk = w_in + (int64_t)E * E;            // W_k block of the packed in_proj_weight
#pragma unroll 8
            for (int j = jg; j < hd; j += 4) acc += qsl[j] * X::to_f32(wk[(int64_t)(h * hd + j) * E + k]);
        }
        red[jg][threadIdx.x & 63] = acc;
        __syncthreads();
        if (jg == 0) {
            const int t = threadIdx.x;
            const float v = red[0][t] + red[1][t] + red[2][t] + red[3][t];
            a_f32[h * E + k] = v;
            const elem hi = X::from_f32(v);
            a_hi[h * E + k] = hi;
            if (a_lo) a_lo[h * E + k] = X::from_f32(v - X::to_f32(hi));
        }
        return;
    }
    id -= nA;
    const int nt = (E / 64) * (E / 64);
    const int ntr = (t_src0 ? nt : 0) + (t_src1 ? nt : 0);
    if (id >= ntr) {
        // fragment-major copies for the weight-stationary kernels (bf16): chunk ((WG*CT + c)*KT + ks)*64 + lane holds the
        // 16 bytes MFMA lane (r16, lg) of the wave owning column group WG loads for column tile c, K-step ks -- so that
        // kernel's weight prologue is 1 KB contiguous per wave-instruction instead of 16 x 64-byte pieces
        id -= ntr;
        const int bpj = (E * E / 8 + 255) / 256;                 // blocks per job
        const int job = id / bpj;
        const int chunk = (id - job * bpj) * 256 + threadIdx.x;
        if (job < fj.n && chunk < E * E / 8) {
            const int CT = E <= 512 ? 2 : 1, KT = E / 32;
            const int lane = chunk & 63, ks = (chunk >> 6) % KT, rest = (chunk >> 6) / KT;
            const int c = rest % CT, WG = rest / CT;
            const int r16 = lane & 15, lg = lane >> 4;
            const int n = CT == 2 ? 32 * WG + 8 * (r16 >> 2) + 4 * c + (r16 & 3) : 16 * WG + r16;
            const int k0 = 32 * ks + 8 * lg;
            const unsigned short* src = reinterpret_cast<const unsigned short*>(fj.src[job]);
            u32x4 v;
            if (!fj.transposed[job]) {
                v = *reinterpret_cast<const u32x4*>(src + (int64_t)n * E + k0);
                reinterpret_cast<u32x4*>(fj.dst[job])[chunk] = v;
            }
        }
        if (job < fj.n && fj.transposed[job]) {                   // (block-uniform: a block lies inside one job)
            // W^T[n][k0 + j] = W[k0 + j][n]: the block's 256 chunks are 4 K-steps (128 rows k) of ONE (WG, c) column set (KT % 4 == 0),
            // i.e. a [128 k][32 n] (CT = 2) / [128][16] piece of W: staged through LDS with 16-byte loads of whole 64- / 32-byte
            // row pieces, each thread then picks its 8 elements out of LDS (round 1-3: eight scattered 2-byte global loads per
            // thread -- most of this launch's time at E = 1024)
            __shared__ __attribute__((aligned(16))) unsigned short ttile[128][40];
            const int CT = E <= 512 ? 2 : 1, KT = E / 32;
            const int chunk0 = (id - job * bpj) * 256;
            const int ks0 = (chunk0 >> 6) % KT, rest0 = (chunk0 >> 6) / KT;
            const int c0 = rest0 % CT, WG0 = rest0 / CT;
            const int nb = CT == 2 ? 32 * WG0 : 16 * WG0;
            const unsigned short* src = reinterpret_cast<const unsigned short*>(fj.src[job]);
            const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
            const unsigned short* gp = src + (int64_t)(32 * ks0 + row) * E + nb + (CT == 2 ? 16 : 8) * half;
            if (CT == 2) {
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(gp), a1 = *reinterpret_cast<const u32x4*>(gp + 8);
                *reinterpret_cast<u32x4*>(&ttile[row][16 * half]) = a0;
                *reinterpret_cast<u32x4*>(&ttile[row][16 * half + 8]) = a1;
            } else {
                *reinterpret_cast<u32x4*>(&ttile[row][8 * half]) = *reinterpret_cast<const u32x4*>(gp);
            }
            __syncthreads();
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r16 = lane & 15, lg = lane >> 4;
            const int nl = CT == 2 ? 8 * (r16 >> 2) + 4 * c0 + (r16 & 3) : r16;
            const int kl = 32 * wv + 8 * lg;
            u32x4 v;
#pragma unroll
            for (int d = 0; d < 4; ++d) v[d] = (unsigned)ttile[kl + 2 * d][nl] | ((unsigned)ttile[kl + 2 * d + 1][nl] << 16);
            reinterpret_cast<u32x4*>(fj.dst[job])[chunk0 + threadIdx.x] = v;
        }
        return;
    }
    // 64 x 64 tiles (E % 64 == 0), whole 16-byte pieces on both global sides: thread = (row t >> 2, quarter t & 3) moves
    // 16 elements of its row in, and 16 elements of its transposed row out (round 1-3: 32 x 32 tiles through 2-byte / 4-byte
    // global accesses -- most of this launch's 26 us at E = 1024)
    const elem* src = id < nt ? t_src0 : t_src1;
    elem* dst = id < nt ? t_dst0 : t_dst1;
    if (id >= nt) id -= nt;
    constexpr int V = 16 / X::BYTES;                               // elements per 16-byte piece
    __shared__ __attribute__((aligned(16))) elem tt[64][64 + V];
    const int row = threadIdx.x >> 2, qt = threadIdx.x & 3;
    const int j0 = (id / (E / 64)) * 64, k0 = (id % (E / 64)) * 64;
    {
        const elem* g = src + (int64_t)(j0 + row) * E + k0 + 16 * qt;
#pragma unroll
        for (int v = 0; v < 16 / V; ++v)
            *reinterpret_cast<u32x4*>(&tt[row][16 * qt + V * v]) = *reinterpret_cast<const u32x4*>(g + V * v);
    }
    __syncthreads();
    {
        __attribute__((aligned(16))) elem o[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = tt[16 * qt + e][row];
        elem* g = dst + (int64_t)(k0 + row) * E + j0 + 16 * qt;
#pragma unroll
        for (int v = 0; v < 16 / V; ++v) *reinterpret_cast<u32x4*>(g + V * v) = *reinterpret_cast<const u32x4*>(&o[V * v]);
    }
}

// ------------------------------------------------------------------------------------------
// gate_fwd: scores via MFMA (x rows as the A operand, A^T as the B operand), softmax over the M
// modalities in registers, head mean by a 16-lane butterfly, curriculum masking per sample.
// One wave = 16 samples; a block = 4 waves.  KSPLIT (small batches: fewer than ~8 waves per CU would otherwise be all
// the memory-level parallelism there is): the block's 4 waves take a quarter of K each for the SAME 16 samples, the
// partial scores are added through LDS in wave order and wave 0 finishes the samples.
template <typename T, int M_, bool KSPLIT>
__global__ __launch_bounds__(256) void gate_fwd_kernel(GateArgs p) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    typedef typename X::frag frag;
    __shared__ float kpart[KSPLIT ? 3 * M_ * 4 * 64 : 1];
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
    const int64_t b0 = KSPLIT ? (int64_t)blockIdx.x * 16 : ((int64_t)blockIdx.x * 4 + wave_id()) * 16;
    if (b0 >= p.B) return;
    const int E = p.E, H = p.H;
    const int64_t brow = (b0 + r16 < p.B) ? (b0 + r16) : (p.B - 1);
    const elem* xrow = reinterpret_cast<const elem*>(p.x) + brow * M_ * (int64_t)E + X::EPL * lg;
    const elem* ahi = reinterpret_cast<const elem*>(p.a_hi) + (int64_t)r16 * E + X::EPL * lg;
    const elem* alo = reinterpret_cast<const elem*>(p.a_lo) + (int64_t)r16 * E + X::EPL * lg;

    f32x4 acc[M_];
#pragma unroll
    for (int m = 0; m < M_; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The score of a (sample, head, modality) is DEFINED as the sum, in order, of its four K-quarter products whenever K
    // splits into quarters of whole K-steps -- so that the KSPLIT form (one quarter per wave) and this one give the same
    // bits, and a sample's result does not depend on the batch it arrives in.
    const bool quarters = E % (4 * X::KSTEP) == 0;
    auto span = [&](int kbeg, int kend, f32x4* a) {
        for (int k0 = kbeg; k0 < kend; k0 += X::KSTEP) {
            frag bh = X::load(ahi + k0);
            frag bl;
            if (X::BYTES == 2) bl = X::load(alo + k0);
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                frag xf = X::load(xrow + (int64_t)m * E + k0);
                a[m] = X::mma(xf, bh, a[m]);
                if (X::BYTES == 2) a[m] = X::mma(xf, bl, a[m]);
            }
        }
    };
    if (KSPLIT) {
        span(wave_id() * (E / 4), (wave_id() + 1) * (E / 4), acc);
    } else if (quarters) {
        span(0, E / 4, acc);
        for (int qk = 1; qk < 4; ++qk) {
            f32x4 part[M_];
#pragma unroll
            for (int m = 0; m < M_; ++m) part[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            span(qk * (E / 4), (qk + 1) * (E / 4), part);
#pragma unroll
            for (int m = 0; m < M_; ++m) acc[m] += part[m];
        }
    } else {
        span(0, E, acc);
    }

    if (KSPLIT) {                                  // waves 1..3 hand their partial scores to wave 0 (fixed order of addition)
        const int w = wave_id();
        if (w > 0) {
#pragma unroll
            for (int m = 0; m < M_; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) kpart[(((w - 1) * M_ + m) * 4 + r) * 64 + lane] = acc[m][r];
        }
        __syncthreads();
        if (w > 0) return;
#pragma unroll
        for (int ww = 0; ww < 3; ++ww)
#pragma unroll
            for (int m = 0; m < M_; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][r] += kpart[((ww * M_ + m) * 4 + r) * 64 + lane];
    }
    // acc[m][r]: sample b0 + 4*lg + r, head r16
    float pr[M_][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t bs = b0 + 4 * lg + r;
        const int64_t bc = bs < p.B ? bs : p.B - 1;
        float sc[M_];
        float mx = -INFINITY;
#pragma unroll
        for (int m = 0; m < M_; ++m) {
            sc[m] = acc[m][r];
            if (p.kpm && p.kpm[bc * M_ + m]) sc[m] = -INFINITY;
            mx = fmaxf(mx, sc[m]);
        }
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < M_; ++m) { sc[m] = expf(sc[m] - mx); sum += sc[m]; }
#pragma unroll
        for (int m = 0; m < M_; ++m) {
            pr[m][r] = sc[m] / sum;
            if (r16 < H && bs < p.B) p.probs[(bs * H + r16) * M_ + m] = pr[m][r];
        }
    }
    // head mean: zero the padded head columns, butterfly over the 16 lanes of the group
    const float invH = 1.0f / (float)H;
    float wsel[M_];
#pragma unroll
    for (int m = 0; m < M_; ++m) {
        wsel[m] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = reduce_r16(r16 < H ? pr[m][r] : 0.f) * invH;
            if (r16 == r) wsel[m] = v;
        }
    }
    if (r16 < 4) {
        const int64_t bs = b0 + 4 * lg + r16;
        if (bs < p.B) {
#pragma unroll
            for (int m = 0; m < M_; ++m) p.attn_w[bs * M_ + m] = wsel[m];
            typedef typename X::elem elem;                 // info copies in the activation dtype
            if (p.i_attn_w) {
#pragma unroll
                for (int m = 0; m < M_; ++m) reinterpret_cast<elem*>(p.i_attn_w)[bs * M_ + m] = X::from_f32(wsel[m]);
            }
            if (p.mask.mode != 0) {
                float w[M_], u[M_], mk[M_];
#pragma unroll
                for (int m = 0; m < M_; ++m) {
                    w[m] = wsel[m];
                    u[m] = p.mask.mode != 1 ? 0.f : (p.uniforms ? p.uniforms[bs * M_ + m] : (p.ph.threads ? philox_uniform_at(p.ph, bs * M_ + m) : 0.f));
                }
                float ent, rate;
                unsigned int bits;
                curriculum_row<M_>(p.mask, M_, w, u, mk, ent, rate, bits);
                if (p.masked_w) {
#pragma unroll
                    for (int m = 0; m < M_; ++m) p.masked_w[bs * M_ + m] = mk[m];
                }
                if (p.entropy) p.entropy[bs] = ent;
                if (p.mask_rate) p.mask_rate[bs] = rate;
                if (p.i_masked_w) {
#pragma unroll
                    for (int m = 0; m < M_; ++m) reinterpret_cast<elem*>(p.i_masked_w)[bs * M_ + m] = X::from_f32(mk[m]);
                }
                if (p.i_entropy) reinterpret_cast<elem*>(p.i_entropy)[bs] = X::from_f32(ent);
                if (p.i_mask_rate) reinterpret_cast<elem*>(p.i_mask_rate)[bs] = X::from_f32(rate);
                if (p.i_target) reinterpret_cast<elem*>(p.i_target)[bs] = X::from_f32(p.target_value);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
#define LAUNCH_CHECKED(...) __VA_ARGS__

void launch_prep_qs(int dtype, const void* w_in, const void* b_in, const void* query, float* qs, int E, float scale,
                    hipStream_t s) {
    dim3 grid((E + 3) / 4), block(256);
    if (dtype == 0)
        prep_qs_kernel<BF16><<<grid, block, 0, s>>>((const unsigned short*)w_in, (const unsigned short*)b_in,
                                                    (const unsigned short*)query, qs, E, scale);
    else
        prep_qs_kernel<F32><<<grid, block, 0, s>>>((const float*)w_in, (const float*)b_in, (const float*)query, qs, E,
                                                   scale);
}

void launch_prep_amat(int dtype, const void* w_in, const float* qs, float* a_f32, void* a_hi, void* a_lo, int E, int H,
                      hipStream_t s) {
    dim3 grid(E / 64, HPAD), block(256);
    if (dtype == 0)
        prep_amat_kernel<BF16><<<grid, block, 0, s>>>((const unsigned short*)w_in, qs, a_f32, (unsigned short*)a_hi,
                                                      (unsigned short*)a_lo, E, H);
    else
        prep_amat_kernel<F32><<<grid, block, 0, s>>>((const float*)w_in, qs, a_f32, (float*)a_hi, (float*)nullptr, E, H);
}

void launch_prep_all(int dtype, const void* w_in, const void* b_in, const void* query, float scale, float* qs, float* a_f32,
                     void* a_hi, void* a_lo, const void* t_src0, void* t_dst0, const void* t_src1, void* t_dst1, int E,
                     int H, const FragJobs& fj, hipStream_t s) {
    const int nt = (E / 64) * (E / 64);
    const int nfrag = dtype == 0 ? fj.n * ((E * E / 8 + 255) / 256) : 0;
    dim3 grid((E / 64) * HPAD + (t_src0 ? nt : 0) + (t_src1 ? nt : 0) + nfrag), block(256);
    if (dtype == 0)
        prep_all_kernel<BF16><<<grid, block, 0, s>>>((const unsigned short*)w_in, (const unsigned short*)b_in,
                                                     (const unsigned short*)query, scale, qs, a_f32, (unsigned short*)a_hi,
                                                     (unsigned short*)a_lo, (const unsigned short*)t_src0,
                                                     (unsigned short*)t_dst0, (const unsigned short*)t_src1,
                                                     (unsigned short*)t_dst1, E, H, fj);
    else
        prep_all_kernel<F32><<<grid, block, 0, s>>>((const float*)w_in, (const float*)b_in, (const float*)query, scale, qs,
                                                    a_f32, (float*)a_hi, (float*)nullptr, (const float*)t_src0,
                                                    (float*)t_dst0, (const float*)t_src1, (float*)t_dst1, E, H, FragJobs{});
}

void launch_transpose(int dtype, const void* src, void* dst, int E, hipStream_t s) {
    dim3 grid((E + 31) / 32, (E + 31) / 32), block(256);
    if (dtype == 0)
        transpose_kernel<BF16><<<grid, block, 0, s>>>((const unsigned short*)src, (unsigned short*)dst, E);
    else
        transpose_kernel<F32><<<grid, block, 0, s>>>((const float*)src, (float*)dst, E);
}

// Per-sample statistics from the per-head softmax weights (used when the scores are produced inside the value-projection
// kernel, aecf_gemm_ws.hip): head mean, curriculum masking, info copies.  One thread per sample; heads summed in order.
template <typename T, int M_>
__global__ __launch_bounds__(256) void gate_stats_kernel(GateArgs p) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    __shared__ float red[4];
    const int64_t bs = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float sq = 0.f;                                              // this row's term of the entropy regulariser
    if (bs < p.B) {
        const int H = p.H;
        float wsel[M_];
#pragma unroll
        for (int m = 0; m < M_; ++m) wsel[m] = 0.f;
        for (int h = 0; h < H; ++h)
#pragma unroll
            for (int m = 0; m < M_; ++m) wsel[m] += p.probs[(bs * H + h) * M_ + m];
        const float invH = 1.0f / (float)H;
#pragma unroll
        for (int m = 0; m < M_; ++m) {
            wsel[m] *= invH;
            p.attn_w[bs * M_ + m] = wsel[m];
            if (p.i_attn_w) reinterpret_cast<elem*>(p.i_attn_w)[bs * M_ + m] = X::from_f32(wsel[m]);
        }
        if (p.mask.mode != 0) {
            float w[M_], u[M_], mk[M_];
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                w[m] = wsel[m];
                u[m] = p.mask.mode != 1 ? 0.f : (p.uniforms ? p.uniforms[bs * M_ + m] : (p.ph.threads ? philox_uniform_at(p.ph, bs * M_ + m) : 0.f));
            }
            float ent, rate;
            unsigned int bits;
            curriculum_row<M_>(p.mask, M_, w, u, mk, ent, rate, bits);
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                if (p.masked_w) p.masked_w[bs * M_ + m] = mk[m];
                if (p.i_masked_w) reinterpret_cast<elem*>(p.i_masked_w)[bs * M_ + m] = X::from_f32(mk[m]);
            }
            if (p.entropy) p.entropy[bs] = ent;
            if (p.mask_rate) p.mask_rate[bs] = rate;
            if (p.i_entropy) reinterpret_cast<elem*>(p.i_entropy)[bs] = X::from_f32(ent);
            if (p.i_mask_rate) reinterpret_cast<elem*>(p.i_mask_rate)[bs] = X::from_f32(rate);
            if (p.i_target) reinterpret_cast<elem*>(p.i_target)[bs] = X::from_f32(p.target_value);
            // CurriculumMasking.entropy_loss (ref :285-314) reads the entropy as the info tensor holds it: its nan_to_num,
            // the difference to the target and the square, summed per block here (the final launch adds the blocks)
            float hv = p.i_entropy ? X::to_f32(X::from_f32(ent)) : ent;
            hv = (hv != hv) ? 0.f : (isinf(hv) ? (hv > 0.f ? 1.f : 0.f) : hv);
            sq = (hv - p.target_value) * (hv - p.target_value);
        }
    }
    if (p.ent_partial) {                                         // (block-uniform)
        sq = reduce_wave(sq);
        if (lane_id() == 0) red[wave_id()] = sq;
        __syncthreads();
        if (threadIdx.x == 0) p.ent_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

void launch_gate_stats(int dtype, const GateArgs& a, hipStream_t s) {
    dim3 grid((unsigned)((a.B + 255) / 256)), block(256);
    AECF_DISPATCH_M(a.M, {
        if (dtype == 0) gate_stats_kernel<BF16, M_><<<grid, block, 0, s>>>(a);
        else gate_stats_kernel<F32, M_><<<grid, block, 0, s>>>(a);
    });
}

void launch_gate_fwd(int dtype, const GateArgs& a, hipStream_t s) {
    // fewer than ~2048 waves of 16 samples (8 per CU): split K over the block's waves instead of the samples
    const int kstep = dtype == 0 ? 32 : 16;
    const bool ksplit = (a.B + 15) / 16 < 2048 && a.E % (4 * kstep) == 0;
    dim3 grid((unsigned)(ksplit ? (a.B + 15) / 16 : (a.B + 63) / 64)), block(256);
    AECF_DISPATCH_M(a.M, {
        if (dtype == 0) {
            if (ksplit) gate_fwd_kernel<BF16, M_, true><<<grid, block, 0, s>>>(a);
            else gate_fwd_kernel<BF16, M_, false><<<grid, block, 0, s>>>(a);
        } else {
            if (ksplit) gate_fwd_kernel<F32, M_, true><<<grid, block, 0, s>>>(a);
            else gate_fwd_kernel<F32, M_, false><<<grid, block, 0, s>>>(a);
        }
    });
}

}  // namespace aecf
