// Stand-alone pieces of the AECF surface: CurriculumMasking forward/backward on free-standing weight
// rows, entropy_loss forward+backward, projection-free single-head attention.  gfx950.
#include <type_traits>
#include "aecf_kernels.h"

namespace aecf {

// ------------------------------------------------------------------------------------------
// CurriculumMasking.forward (ref aecf/AECFLayer.py:130-283): one thread per row
template <int LMAX>
__global__ __launch_bounds__(256) void mask_fwd_kernel(int64_t rows, int L, MaskCfg cfg, const float* __restrict__ w,
                                                       const float* __restrict__ u, float* __restrict__ masked,
                                                       float* __restrict__ entropy, float* __restrict__ mask_rate,
                                                       uint8_t* __restrict__ bits_out) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float wv[LMAX], uv[LMAX], mk[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) {
        wv[i] = i < L ? w[row * L + i] : 0.f;
        uv[i] = (i < L && u) ? u[row * L + i] : 0.f;
        mk[i] = 0.f;
    }
    float ent, rate;
    typedef typename std::conditional<(LMAX > 32), unsigned long long, unsigned int>::type bits_t;
    bits_t bits;
    curriculum_row<LMAX, bits_t>(cfg, L, wv, uv, mk, ent, rate, bits);
#pragma unroll
    for (int i = 0; i < LMAX; ++i)
        if (i < L) {
            if (masked) masked[row * L + i] = mk[i];
            if (bits_out) bits_out[row * L + i] = (uint8_t)((bits >> i) & 1);
        }
    if (entropy) entropy[row] = ent;
    if (mask_rate) mask_rate[row] = rate;
}

// gradient of the stand-alone module's outputs w.r.t. its input weights.
// train: final = valid ? (w_n * mask) / sum(w_n * mask) : w_n, w_n = w / sum(w)   (entropy detached, ref :278)
// eval : masked = w (identity), entropy = clamp(-sum xlogy(w,w))
template <int LMAX>
__global__ __launch_bounds__(256) void mask_bwd_kernel(int64_t rows, int L, int mode, float eps, float log_L,
                                                       const float* __restrict__ w, const uint8_t* __restrict__ bits,
                                                       const float* __restrict__ d_masked,
                                                       const float* __restrict__ d_entropy, float* __restrict__ d_w) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float wv[LMAX], g[LMAX], out[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; ++i) {
        wv[i] = i < L ? w[row * L + i] : 0.f;
        g[i] = (i < L && d_masked) ? d_masked[row * L + i] : 0.f;
        out[i] = 0.f;
    }
    if (mode == 2) {
        float h = 0.f;
#pragma unroll
        for (int i = 0; i < LMAX; ++i)
            if (i < L) h -= xlogx(wv[i]);
        const bool live = (h >= 0.f) && (h <= log_L);
        const float de = d_entropy ? d_entropy[row] : 0.f;
#pragma unroll
        for (int i = 0; i < LMAX; ++i)
            if (i < L) out[i] = g[i] + ((live && d_entropy) ? -(logf(wv[i]) + 1.0f) * de : 0.f);
    } else if (L <= 1) {
        out[0] = g[0];
    } else {
        bool fin[LMAX];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LMAX; ++i) {
            fin[i] = isfinite(wv[i]);
            if (!fin[i]) wv[i] = 0.f;
            if (i < L) s += wv[i];
        }
        if (!(s < eps)) {
            float wn[LMAX], mk[LMAX];
            float ms = 0.f;
#pragma unroll
            for (int i = 0; i < LMAX; ++i) {
                wn[i] = wv[i] / s;
                mk[i] = (i < L && bits[row * L + i]) ? 1.f : 0.f;
                ms += wn[i] * mk[i];
            }
            float dwn[LMAX];
            if (ms > eps) {
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < LMAX; ++i) dot += g[i] * (wn[i] * mk[i] / ms);
#pragma unroll
                for (int i = 0; i < LMAX; ++i) dwn[i] = mk[i] * (g[i] - dot) / ms;
            } else {
#pragma unroll
                for (int i = 0; i < LMAX; ++i) dwn[i] = g[i];
            }
            float dot2 = 0.f;
#pragma unroll
            for (int i = 0; i < LMAX; ++i) dot2 += dwn[i] * wn[i];
#pragma unroll
            for (int i = 0; i < LMAX; ++i) out[i] = fin[i] ? (dwn[i] - dot2) / s : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < LMAX; ++i)
        if (i < L) d_w[row * L + i] = out[i];
}

// ------------------------------------------------------------------------------------------
// The same two functions for rows of ANY length (the reference's module is length-agnostic, ref :130-283): one wave per row,
// key i belongs to lane i % 64 throughout, every sum is the lanes' partial sums (ascending i) folded by the wave butterfly.
// The keep bits live in the caller's mask_bits bytes (required in train mode); a lane only ever reads back bytes it wrote.
__global__ __launch_bounds__(256) void mask_fwd_long_kernel(int64_t rows, int L, MaskCfg c, const float* __restrict__ w,
                                                            const float* __restrict__ u, float* __restrict__ masked,
                                                            float* __restrict__ entropy, float* __restrict__ mask_rate,
                                                            uint8_t* __restrict__ bits) {
    const int64_t row = (int64_t)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;                                      // (whole waves)
    const int lane = lane_id();
    const float* wr = w + row * L;
    if (c.mode == 2) {                                            // eval (ref :150-156)
        float h = 0.f;
        for (int i = lane; i < L; i += 64) {
            h -= xlogx(wr[i]);
            if (masked) masked[row * L + i] = wr[i];
            if (bits) bits[row * L + i] = 1;
        }
        h = reduce_wave(h);
        float e = fminf(fmaxf(h, 0.f), c.log_L);
        if (h != h) e = h;
        if (lane == 0) {
            if (entropy) entropy[row] = e;
            if (mask_rate) mask_rate[row] = 0.f;
        }
        return;
    }
    // ref :170-184
    float s = 0.f;
    for (int i = lane; i < L; i += 64) { const float v = wr[i]; s += isfinite(v) ? v : 0.f; }
    s = reduce_wave(s);
    const bool needs_norm = s < c.eps;
    auto wn = [&](int i) -> float { const float v = wr[i]; return needs_norm ? c.inv_L : ((isfinite(v) ? v : 0.f) / s); };
    // ref :190-201
    float h = 0.f;
    for (int i = lane; i < L; i += 64) h -= xlogx(wn(i));
    h = fminf(fmaxf(reduce_wave(h), 0.f), c.log_L);
    const float ne = fminf(fmaxf(h / c.log_L, 0.f), 1.f);
    const float keep = fminf(fmaxf(1.0f - c.base_mask_prob * ne, 0.f), 1.f);
    // ref :204
    uint8_t* br = bits + row * L;
    float act = 0.f;
    for (int i = lane; i < L; i += 64) {
        const uint8_t b = u[row * L + i] < keep ? 1 : 0;
        br[i] = b;
        act += (float)b;
    }
    int active = (int)(reduce_wave(act) + 0.5f);                  // (exact: integers below 2^24)
    // ref :207-260: rows with too few survivors keep exactly their top-k weights (lowest index on ties)
    const int k = c.min_active < L ? c.min_active : L;
    if (active < k) {
        for (int i = lane; i < L; i += 64) br[i] = 0;
        for (int t = 0; t < k; ++t) {
            float bv = -INFINITY;
            int best = 0x7fffffff;
            for (int i = lane; i < L; i += 64) {
                const float v = wn(i);
                if (!br[i] && (best == 0x7fffffff || v > bv)) { bv = v; best = i; }
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float ov = __shfl_xor(bv, off, 64);
                const int oi = __shfl_xor(best, off, 64);
                if (oi != 0x7fffffff && (best == 0x7fffffff || ov > bv || (ov == bv && oi < best))) { bv = ov; best = oi; }
            }
            if ((best & 63) == lane) br[best] = 1;
        }
        active = k;
    }
    // ref :263-272
    float ms = 0.f;
    for (int i = lane; i < L; i += 64) ms += br[i] ? wn(i) : 0.f;
    ms = reduce_wave(ms);
    const bool valid = ms > c.eps;
    if (masked)
        for (int i = lane; i < L; i += 64) {
            const float v = wn(i);
            masked[row * L + i] = valid ? ((br[i] ? v : 0.f) / ms) : v;
        }
    if (lane == 0) {
        if (entropy) entropy[row] = h;
        if (mask_rate) mask_rate[row] = 1.0f - (float)active / (float)L;     // ref :275
    }
}

__global__ __launch_bounds__(256) void mask_bwd_long_kernel(int64_t rows, int L, int mode, float eps, float log_L,
                                                            const float* __restrict__ w, const uint8_t* __restrict__ bits,
                                                            const float* __restrict__ d_masked,
                                                            const float* __restrict__ d_entropy, float* __restrict__ d_w) {
    const int64_t row = (int64_t)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    const int lane = lane_id();
    const float* wr = w + row * L;
    const float* gr = d_masked ? d_masked + row * L : nullptr;
    float* out = d_w + row * L;
    if (mode == 2) {
        float h = 0.f;
        for (int i = lane; i < L; i += 64) h -= xlogx(wr[i]);
        h = reduce_wave(h);
        const bool live = (h >= 0.f) && (h <= log_L);
        const float de = d_entropy ? d_entropy[row] : 0.f;
        for (int i = lane; i < L; i += 64)
            out[i] = (gr ? gr[i] : 0.f) + ((live && d_entropy) ? -(logf(wr[i]) + 1.0f) * de : 0.f);
        return;
    }
    float s = 0.f;
    for (int i = lane; i < L; i += 64) { const float v = wr[i]; s += isfinite(v) ? v : 0.f; }
    s = reduce_wave(s);
    if (s < eps) {                                                // uniform fallback: no dependence on the weights
        for (int i = lane; i < L; i += 64) out[i] = 0.f;
        return;
    }
    const uint8_t* br = bits + row * L;
    auto wn = [&](int i) -> float { const float v = wr[i]; return (isfinite(v) ? v : 0.f) / s; };
    float ms = 0.f;
    for (int i = lane; i < L; i += 64) ms += br[i] ? wn(i) : 0.f;
    ms = reduce_wave(ms);
    const bool valid = ms > eps;
    float dot = 0.f;
    if (valid) {
        for (int i = lane; i < L; i += 64) dot += br[i] ? (gr ? gr[i] : 0.f) * (wn(i) / ms) : 0.f;
        dot = reduce_wave(dot);
    }
    auto dwn = [&](int i) -> float {
        const float g = gr ? gr[i] : 0.f;
        return valid ? (br[i] ? (g - dot) / ms : 0.f) : g;
    };
    float dot2 = 0.f;
    for (int i = lane; i < L; i += 64) dot2 += dwn(i) * wn(i);
    dot2 = reduce_wave(dot2);
    for (int i = lane; i < L; i += 64) out[i] = isfinite(wr[i]) ? (dwn(i) - dot2) / s : 0.f;
}

// ------------------------------------------------------------------------------------------
// entropy_loss (ref aecf/AECFLayer.py:285-314): mean((nan_to_num(H) - target)^2), two-stage reduce
__device__ __forceinline__ float nan_to_num_ref(float e) {   // nan=0, +inf=1, -inf=0 (ref :296)
    if (e != e) return 0.f;
    if (isinf(e)) return e > 0.f ? 1.f : 0.f;
    return e;
}

template <typename T>
__global__ __launch_bounds__(256) void entropy_loss_partial_kernel(int64_t n, float target,
                                                                   const typename Tr<T>::elem* __restrict__ e,
                                                                   float scale_grad, float* __restrict__ d_e,
                                                                   float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float raw = Tr<T>::to_f32(e[i]);
        const float d = nan_to_num_ref(raw) - target;
        acc += d * d;
        if (d_e) d_e[i] = isfinite(raw) ? scale_grad * d : 0.f;
    }
    acc = reduce_wave(acc);
    if (lane_id() == 0) red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(256) void entropy_loss_final_kernel(int nblk, float inv_n, const float* __restrict__ partial,
                                                                 typename Tr<T>::elem* __restrict__ loss) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) acc += partial[i];
    acc = reduce_wave(acc);
    if (lane_id() == 0) red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = Tr<T>::from_f32(fmaxf((red[0] + red[1] + red[2] + red[3]) * inv_n, 0.f));
}

// ------------------------------------------------------------------------------------------
// projection-free attention (ref aecf/AECFLayer.py:556-581).  One block per batch element; S, T <= 64.
template <typename T>
__global__ __launch_bounds__(256) void sdpa_fwd_kernel(int S, int Tn, int E, float scale,
                                                       const typename Tr<T>::elem* __restrict__ q,
                                                       const typename Tr<T>::elem* __restrict__ k,
                                                       const typename Tr<T>::elem* __restrict__ v,
                                                       typename Tr<T>::elem* __restrict__ out, float* __restrict__ probs) {
    using X = Tr<T>;
    __shared__ float sc[64][65];
    const int64_t b = blockIdx.x;
    const int lane = lane_id(), w = wave_id();
    q += b * S * (int64_t)E; k += b * Tn * (int64_t)E; v += b * Tn * (int64_t)E; out += b * S * (int64_t)E;
    for (int pair = w; pair < S * Tn; pair += 4) {
        const int s = pair / Tn, t = pair % Tn;
        float a = 0.f;
        for (int e = lane; e < E; e += 64) a += X::to_f32(q[(int64_t)s * E + e]) * X::to_f32(k[(int64_t)t * E + e]);
        a = reduce_wave(a);
        if (lane == 0) sc[s][t] = a * scale;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 256) {
        float mx = -INFINITY;
        for (int t = 0; t < Tn; ++t) mx = fmaxf(mx, sc[s][t]);
        float sum = 0.f;
        for (int t = 0; t < Tn; ++t) { float ex = expf(sc[s][t] - mx); sc[s][t] = ex; sum += ex; }
        for (int t = 0; t < Tn; ++t) { float pv = sc[s][t] / sum; sc[s][t] = pv; if (probs) probs[(b * S + s) * Tn + t] = pv; }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < S * E; idx += 256) {
        const int s = idx / E, e = idx % E;
        float a = 0.f;
        for (int t = 0; t < Tn; ++t) a += sc[s][t] * X::to_f32(v[(int64_t)t * E + e]);
        out[idx] = X::from_f32(a);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sdpa_bwd_kernel(int S, int Tn, int E, float scale,
                                                       const typename Tr<T>::elem* __restrict__ q,
                                                       const typename Tr<T>::elem* __restrict__ k,
                                                       const typename Tr<T>::elem* __restrict__ v,
                                                       const float* __restrict__ probs,
                                                       const typename Tr<T>::elem* __restrict__ dout,
                                                       typename Tr<T>::elem* __restrict__ dq,
                                                       typename Tr<T>::elem* __restrict__ dk,
                                                       typename Tr<T>::elem* __restrict__ dv) {
    using X = Tr<T>;
    __shared__ float pp[64][65];
    __shared__ float ds[64][65];
    const int64_t b = blockIdx.x;
    const int lane = lane_id(), w = wave_id();
    q += b * S * (int64_t)E; k += b * Tn * (int64_t)E; v += b * Tn * (int64_t)E; dout += b * S * (int64_t)E;
    dq += b * S * (int64_t)E; dk += b * Tn * (int64_t)E; dv += b * Tn * (int64_t)E;
    for (int pair = w; pair < S * Tn; pair += 4) {
        const int s = pair / Tn, t = pair % Tn;
        float a = 0.f;
        for (int e = lane; e < E; e += 64) a += X::to_f32(dout[(int64_t)s * E + e]) * X::to_f32(v[(int64_t)t * E + e]);
        a = reduce_wave(a);
        if (lane == 0) { ds[s][t] = a; pp[s][t] = probs[(b * S + s) * Tn + t]; }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 256) {
        float dot = 0.f;
        for (int t = 0; t < Tn; ++t) dot += pp[s][t] * ds[s][t];
        for (int t = 0; t < Tn; ++t) ds[s][t] = pp[s][t] * (ds[s][t] - dot) * scale;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < S * E; idx += 256) {
        const int s = idx / E, e = idx % E;
        float a = 0.f;
        for (int t = 0; t < Tn; ++t) a += ds[s][t] * X::to_f32(k[(int64_t)t * E + e]);
        dq[idx] = X::from_f32(a);
    }
    for (int idx = threadIdx.x; idx < Tn * E; idx += 256) {
        const int t = idx / E, e = idx % E;
        float a = 0.f, c = 0.f;
        for (int s = 0; s < S; ++s) {
            a += ds[s][t] * X::to_f32(q[(int64_t)s * E + e]);
            c += pp[s][t] * X::to_f32(dout[(int64_t)s * E + e]);
        }
        dk[idx] = X::from_f32(a);
        dv[idx] = X::from_f32(c);
    }
}

// ------------------------------------------------------------------------------------------
void launch_mask_fwd(int64_t rows, int L, const MaskCfg& cfg, const float* w, const float* u, float* masked,
                     float* entropy, float* mask_rate, uint8_t* bits, hipStream_t s) {
    dim3 grid((unsigned)((rows + 255) / 256)), block(256);
    if (L <= 4) mask_fwd_kernel<4><<<grid, block, 0, s>>>(rows, L, cfg, w, u, masked, entropy, mask_rate, bits);
    else if (L <= 8) mask_fwd_kernel<8><<<grid, block, 0, s>>>(rows, L, cfg, w, u, masked, entropy, mask_rate, bits);
    else if (L <= 16) mask_fwd_kernel<16><<<grid, block, 0, s>>>(rows, L, cfg, w, u, masked, entropy, mask_rate, bits);
    else if (L <= 32) mask_fwd_kernel<32><<<grid, block, 0, s>>>(rows, L, cfg, w, u, masked, entropy, mask_rate, bits);
    else if (L <= 64) mask_fwd_kernel<64><<<grid, block, 0, s>>>(rows, L, cfg, w, u, masked, entropy, mask_rate, bits);     // 64-bit keep word
    else mask_fwd_long_kernel<<<dim3((unsigned)((rows + 3) / 4)), block, 0, s>>>(rows, L, cfg, w, u, masked, entropy, mask_rate, bits);
}

void launch_mask_bwd(int64_t rows, int L, int mode, float eps, float log_L, const float* w, const uint8_t* bits,
                     const float* d_masked, const float* d_entropy, float* d_w, hipStream_t s) {
    dim3 grid((unsigned)((rows + 255) / 256)), block(256);
    if (L <= 4) mask_bwd_kernel<4><<<grid, block, 0, s>>>(rows, L, mode, eps, log_L, w, bits, d_masked, d_entropy, d_w);
    else if (L <= 8) mask_bwd_kernel<8><<<grid, block, 0, s>>>(rows, L, mode, eps, log_L, w, bits, d_masked, d_entropy, d_w);
    else if (L <= 16) mask_bwd_kernel<16><<<grid, block, 0, s>>>(rows, L, mode, eps, log_L, w, bits, d_masked, d_entropy, d_w);
    else if (L <= 32) mask_bwd_kernel<32><<<grid, block, 0, s>>>(rows, L, mode, eps, log_L, w, bits, d_masked, d_entropy, d_w);
    else if (L <= 64) mask_bwd_kernel<64><<<grid, block, 0, s>>>(rows, L, mode, eps, log_L, w, bits, d_masked, d_entropy, d_w);
    else mask_bwd_long_kernel<<<dim3((unsigned)((rows + 3) / 4)), block, 0, s>>>(rows, L, mode, eps, log_L, w, bits, d_masked, d_entropy, d_w);
}

// ------------------------------------------------------------------------------------------
// missing-modality front-end (ref xrays/train_xrays_example.py:156-177, 202-203): one wave per feature row; the row is
// read once, its squared norm reduced in float32, and written zeroed or as is.
template <typename T>
__global__ __launch_bounds__(256) void modality_frontend_kernel(int64_t rows, int dim, const typename Tr<T>::elem* __restrict__ feat,
                                                                const uint8_t* __restrict__ drop,
                                                                typename Tr<T>::elem* out, uint8_t* __restrict__ present) {
    using X = Tr<T>;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave_id();
    if (r >= rows) return;
    const int lane = lane_id();
    const bool dropped = drop && drop[r] != 0;
    const typename X::elem* src = feat + r * dim;
    typename X::elem* dst = out ? out + r * dim : nullptr;
    float ss = 0.f;
    const bool vec = (dim % X::EPL == 0) && ((reinterpret_cast<uintptr_t>(feat) | (out ? reinterpret_cast<uintptr_t>(out) : 0)) % 16 == 0);
    if (vec) {
        for (int c = lane * X::EPL; c < dim; c += 64 * X::EPL) {
            typename X::frag v = X::load(src + c);
            if (dropped) v = X::zero();
            float f[X::EPL];
            X::unpack(v, f);
#pragma unroll
            for (int e = 0; e < X::EPL; ++e) ss = fmaf(f[e], f[e], ss);
            if (dst && (dropped || dst != src)) *reinterpret_cast<typename X::frag*>(dst + c) = v;
        }
    } else {
        for (int c = lane; c < dim; c += 64) {
            const typename X::elem v = dropped ? X::from_f32(0.f) : src[c];
            const float f = X::to_f32(v);
            ss = fmaf(f, f, ss);
            if (dst && (dropped || dst != src)) dst[c] = v;
        }
    }
    ss = reduce_wave(ss);
    if (lane == 0) present[r] = sqrtf(ss) > 1e-6f ? 1 : 0;
}

// bf16 -> float32, n elements (precise mode: the intermediates' consumers run on the float32 kernels)
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float v[8];
        Tr<BF16>::unpack(*reinterpret_cast<const u32x4*>(src + i), v);
        *reinterpret_cast<f32x4*>(dst + i) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dst + i + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = Tr<BF16>::to_f32(src[j]);
    }
}

// the in-kernel uniform draw on its own (aecf_philox_uniforms: tests against torch.rand, callers that want the tensor)
__global__ __launch_bounds__(256) void philox_uniforms_kernel(int64_t n, PhiloxDraw ph, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = philox_uniform_at(ph, i);
}

void launch_philox_uniforms(int64_t n, const PhiloxDraw& ph, float* out, hipStream_t s) {
    philox_uniforms_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(n, ph, out);
}

// float32 -> bf16 (round to nearest even) of up to 8 tensors in ONE launch: the master-weight casts of a mixed-precision step
// (blocks of 2048 elements dealt over the tensors in order; torch's multi-tensor copy hands a block 65536 elements: 21 us for
// the 1 M parameters of the d = 512 layer against ~4 us here)
struct CastJobs {
    int n = 0;
    const float* src[8];
    unsigned short* dst[8];
    int64_t numel[8];
    int64_t first_block[9];
};
__global__ __launch_bounds__(256) void cast_f32_bf16_multi_kernel(CastJobs j) {
    int t = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < j.n && (int64_t)blockIdx.x >= j.first_block[i]) t = i;
    const int64_t i0 = (((int64_t)blockIdx.x - j.first_block[t]) * 256 + threadIdx.x) * 8;
    const float* src = j.src[t];
    unsigned short* dst = j.dst[t];
    const int64_t n = j.numel[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (vec && i0 + 8 <= n) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + i0), b = *reinterpret_cast<const f32x4*>(src + i0 + 4);
        *reinterpret_cast<u32x4*>(dst + i0) = u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]),
                                                    pack_bf16x2(b[2], b[3])};
    } else {
        for (int64_t k = i0; k < n && k < i0 + 8; ++k) dst[k] = Tr<BF16>::from_f32(src[k]);
    }
}

int launch_cast_f32_bf16_multi(int n, const float* const* src, void* const* dst, const int64_t* numel, hipStream_t s) {
    CastJobs j;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        j.src[j.n] = src[i]; j.dst[j.n] = (unsigned short*)dst[i]; j.numel[j.n] = numel[i]; j.first_block[j.n] = blocks;
        if (numel[i] > 0) { blocks += (numel[i] + 2047) / 2048; ++j.n; }
    }
    if (blocks == 0) return 0;
    cast_f32_bf16_multi_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(j);
    return 1;
}

void launch_cast_bf16_f32(const void* src, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0 || !src) return;
    cast_bf16_f32_kernel<<<dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, s>>>((const unsigned short*)src, dst, n);
}

void launch_modality_frontend(int dtype, int64_t rows, int dim, const void* feat, const uint8_t* drop, void* out,
                              uint8_t* present, hipStream_t s) {
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == 0)
        modality_frontend_kernel<BF16><<<grid, block, 0, s>>>(rows, dim, (const unsigned short*)feat, drop,
                                                              (unsigned short*)out, present);
    else
        modality_frontend_kernel<F32><<<grid, block, 0, s>>>(rows, dim, (const float*)feat, drop, (float*)out, present);
}

// small n: ONE block does both stages (a second launch and the gap in front of it cost more than the work); 16-byte loads.
template <typename T>
__global__ __launch_bounds__(1024) void entropy_loss_single_kernel(int64_t n, float target, const typename Tr<T>::elem* __restrict__ e,
                                                                   float scale_grad, float inv_n, float* __restrict__ d_e,
                                                                   typename Tr<T>::elem* __restrict__ loss) {
    constexpr int V = 16 / Tr<T>::BYTES;                         // elements per 16-byte load
    __shared__ float red[16];
    float acc = 0.f;
    const bool vec = (reinterpret_cast<uintptr_t>(e) & 15) == 0;
    const int64_t nv = vec ? n / V : 0;
    for (int64_t c = threadIdx.x; c < nv; c += 1024) {
        float v[V];
        Tr<T>::unpack(*reinterpret_cast<const typename Tr<T>::frag*>(e + c * V), v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = nan_to_num_ref(v[k]) - target;
            acc += d * d;
            if (d_e) d_e[c * V + k] = isfinite(v[k]) ? scale_grad * d : 0.f;
        }
    }
    for (int64_t i = nv * V + threadIdx.x; i < n; i += 1024) {
        const float raw = Tr<T>::to_f32(e[i]);
        const float d = nan_to_num_ref(raw) - target;
        acc += d * d;
        if (d_e) d_e[i] = isfinite(raw) ? scale_grad * d : 0.f;
    }
    acc = reduce_wave(acc);
    if (lane_id() == 0) red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        loss[0] = Tr<T>::from_f32(fmaxf(t * inv_n, 0.f));
    }
}

void launch_entropy_partials(int dtype, int64_t n, float target, const void* entropy, float* partial, hipStream_t s) {
    const int nblk = (int)((n + 255) / 256);                // (one block per 256 rows: the partition gate_stats uses)
    if (dtype == 0)
        entropy_loss_partial_kernel<BF16><<<dim3(nblk), dim3(256), 0, s>>>(n, target, (const unsigned short*)entropy, 0.f, nullptr, partial);
    else
        entropy_loss_partial_kernel<F32><<<dim3(nblk), dim3(256), 0, s>>>(n, target, (const float*)entropy, 0.f, nullptr, partial);
}

void launch_entropy_from_partials(int dtype, int64_t n, const float* partial, void* loss, hipStream_t s) {
    const int nblk = (int)((n + 255) / 256);
    if (dtype == 0) entropy_loss_final_kernel<BF16><<<dim3(1), dim3(256), 0, s>>>(nblk, 1.0f / (float)n, partial, (unsigned short*)loss);
    else entropy_loss_final_kernel<F32><<<dim3(1), dim3(256), 0, s>>>(nblk, 1.0f / (float)n, partial, (float*)loss);
}

void launch_entropy_loss(int dtype, int64_t n, float target, const void* entropy, float upstream, void* loss,
                         float* d_entropy, float* partial, hipStream_t s) {
    if (n <= 8192) {       // (one block: measured 31 us at 65536 rows against 4.6 + 4.3 us for the two-kernel form below)
        const float inv = 1.0f / (float)n;
        if (dtype == 0)
            entropy_loss_single_kernel<BF16><<<dim3(1), dim3(1024), 0, s>>>(n, target, (const unsigned short*)entropy,
                                                                          2.0f * inv * upstream, inv, d_entropy, (unsigned short*)loss);
        else
            entropy_loss_single_kernel<F32><<<dim3(1), dim3(1024), 0, s>>>(n, target, (const float*)entropy,
                                                                         2.0f * inv * upstream, inv, d_entropy, (float*)loss);
        return;
    }
    int nblk = (int)((n + 255) / 256);
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    const float inv_n = 1.0f / (float)n;
    if (dtype == 0) {
        entropy_loss_partial_kernel<BF16><<<dim3(nblk), dim3(256), 0, s>>>(n, target, (const unsigned short*)entropy,
                                                                          2.0f * inv_n * upstream, d_entropy, partial);
        entropy_loss_final_kernel<BF16><<<dim3(1), dim3(256), 0, s>>>(nblk, inv_n, partial, (unsigned short*)loss);
    } else {
        entropy_loss_partial_kernel<F32><<<dim3(nblk), dim3(256), 0, s>>>(n, target, (const float*)entropy,
                                                                         2.0f * inv_n * upstream, d_entropy, partial);
        entropy_loss_final_kernel<F32><<<dim3(1), dim3(256), 0, s>>>(nblk, inv_n, partial, (float*)loss);
    }
}

void launch_sdpa_fwd(int dtype, int64_t B, int S, int T, int E, float scale, const void* q, const void* k, const void* v,
                     void* out, float* probs, hipStream_t s) {
    dim3 grid((unsigned)B), block(256);
    if (dtype == 0)
        sdpa_fwd_kernel<BF16><<<grid, block, 0, s>>>(S, T, E, scale, (const unsigned short*)q, (const unsigned short*)k,
                                                     (const unsigned short*)v, (unsigned short*)out, probs);
    else
        sdpa_fwd_kernel<F32><<<grid, block, 0, s>>>(S, T, E, scale, (const float*)q, (const float*)k, (const float*)v,
                                                    (float*)out, probs);
}

void launch_sdpa_bwd(int dtype, int64_t B, int S, int T, int E, float scale, const void* q, const void* k, const void* v,
                     const float* probs, const void* dout, void* dq, void* dk, void* dv, hipStream_t s) {
    dim3 grid((unsigned)B), block(256);
    if (dtype == 0)
        sdpa_bwd_kernel<BF16><<<grid, block, 0, s>>>(S, T, E, scale, (const unsigned short*)q, (const unsigned short*)k,
                                                     (const unsigned short*)v, probs, (const unsigned short*)dout,
                                                     (unsigned short*)dq, (unsigned short*)dk, (unsigned short*)dv);
    else
        sdpa_bwd_kernel<F32><<<grid, block, 0, s>>>(S, T, E, scale, (const float*)q, (const float*)k, (const float*)v, probs,
                                                    (const float*)dout, (float*)dq, (float*)dk, (float*)dv);
}

}  // namespace aecf
