// Score gradient from the SAVED per-modality value projections (gfx950):
//     da[b,h,m] = do_h[b] . V_h[b,m]        V[b,m,:] = W_v x[b,m,:] + b_v   (written by the forward V projection)
//     dp = da + dwbar[b,m]/H ;  ds[b,h,:] = a * (dp - sum_m a dp)            (softmax backward)
// Memory-bound: one wave per sample reads do[b,:] and the M rows V[b,m,:] once with 16-byte loads; a lane's chunk
// (8 bf16 / 4 f32 consecutive columns) lies inside one head, a head spans LPH = head_dim*BYTES/16 consecutive lanes and
// each wave pass covers 64/LPH whole heads, so the per-head dot is an xor-shuffle tree (LPH a power of two) or a
// fixed-order gather over the head's lanes (any LPH, e.g. head_dim 96) -- deterministic, no LDS, no atomics.
#include "aecf_kernels.h"

namespace aecf {

template <typename T, int M_>
__global__ __launch_bounds__(256) void dscore_v_kernel(BwdGArgs p, const typename Tr<T>::elem* __restrict__ V, int lph) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    constexpr int CH = X::EPL;                        // elements per 16-byte chunk
    const int64_t b = (int64_t)blockIdx.x * 4 + wave_id();
    if (b >= p.B) return;
    const int lane = lane_id();
    const int E = p.E, H = p.H;
    const int hpp = 64 / lph;                         // whole heads per wave pass
    const int hl = lane / lph, within = lane - hl * lph;
    const bool pow2 = (lph & (lph - 1)) == 0;
    const elem* dorow = reinterpret_cast<const elem*>(p.dobuf) + b * (int64_t)E;
    const elem* vrow = V + b * M_ * (int64_t)E;
    const float invH = 1.0f / (float)H;

    float dwb[M_];
#pragma unroll
    for (int m = 0; m < M_; ++m) dwb[m] = p.d_attn_w ? p.d_attn_w[b * M_ + m] : 0.f;
    if (p.d_entropy) {          // eval mode: the entropy keeps its graph (ref :150-156)
        float wv[M_], hsum = 0.f;
#pragma unroll
        for (int m = 0; m < M_; ++m) { wv[m] = p.attn_w[b * M_ + m]; hsum -= xlogx(wv[m]); }
        const bool live = (hsum >= 0.f) && (hsum <= p.log_M);
        const float de = p.d_entropy[b];
#pragma unroll
        for (int m = 0; m < M_; ++m) dwb[m] += live ? -(logf(wv[m]) + 1.0f) * de : 0.f;
    }

    for (int h0 = 0; h0 < H; h0 += hpp) {
        const int h = h0 + hl;
        const bool on = hl < hpp && h < H;
        const int c = h * lph + within;               // this lane's 16-byte chunk of the row
        float part[M_];
#pragma unroll
        for (int m = 0; m < M_; ++m) part[m] = 0.f;
        if (on) {
            float dv[CH];
            X::unpack(X::load(dorow + (int64_t)c * CH), dv);
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                float vv[CH];
                X::unpack(X::load(vrow + (int64_t)m * E + (int64_t)c * CH), vv);
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < CH; ++j) a = fmaf(dv[j], vv[j], a);
                part[m] = a;
            }
        }
        // sum over the lph lanes of each head (heads are lane-aligned: lanes hl*lph .. hl*lph + lph - 1)
        if (pow2) {
            for (int off = 1; off < lph; off <<= 1) {
#pragma unroll
                for (int m = 0; m < M_; ++m) part[m] += __shfl_xor(part[m], off, 64);
            }
        } else {
            float tot[M_];
#pragma unroll
            for (int m = 0; m < M_; ++m) tot[m] = 0.f;
            const int base = hl * lph < 64 ? hl * lph : 0;
            for (int i = 0; i < lph; ++i) {
                const int srcl = (base + i) & 63;
#pragma unroll
                for (int m = 0; m < M_; ++m) tot[m] += __shfl(part[m], srcl, 64);
            }
#pragma unroll
            for (int m = 0; m < M_; ++m) part[m] = tot[m];
        }
        if (on && within == 0) {
            float pm[M_], dp[M_], dot = 0.f;
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                pm[m] = p.probs[(b * H + h) * M_ + m];
                dp[m] = part[m] + dwb[m] * invH;
                dot = fmaf(pm[m], dp[m], dot);
            }
#pragma unroll
            for (int m = 0; m < M_; ++m) p.dsbuf[(b * H + h) * M_ + m] = pm[m] * (dp[m] - dot);
        }
    }
}

bool launch_dscore_v(int dtype, const BwdGArgs& a, const void* saved_v, hipStream_t s) {
    const int bytes = dtype == 0 ? 2 : 4;
    const int lph = a.hd * bytes / 16;               // lanes per head
    if (lph < 1 || lph > 64 || (a.hd * bytes) % 16 != 0) return false;
    dim3 grid((unsigned)((a.B + 3) / 4)), block(256);
    AECF_DISPATCH_M(a.M, {
        if (dtype == 0) dscore_v_kernel<BF16, M_><<<grid, block, 0, s>>>(a, (const unsigned short*)saved_v, lph);
        else dscore_v_kernel<F32, M_><<<grid, block, 0, s>>>(a, (const float*)saved_v, lph);
    });
    return true;
}

}  // namespace aecf
