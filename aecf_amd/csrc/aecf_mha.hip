// General multi-head attention: the options of nn.MultiheadAttention that the shared-query hot
// path does not take -- per-sample queries, tgt_len > 1, key != value, attn_mask, key_padding_mask, dropout.
// Projections and weight gradients reuse the library's GEMMs (gemm_nt / gemm_tn and their bf16 fast forms); this
// file adds the per-sample attention core (forward and backward, one block per sample, all heads) and the C entry
// points.  gfx950.
#include <math.h>
#include <stdint.h>

#include "../../include/aecf_hip.h"
#include "aecf_kernels.h"

namespace aecf {

namespace {

constexpr int LMAX = 4096;            // tgt_len, src_len (the score rows of a query chunk live in LDS)
constexpr int CORE_LDS = 96 * 1024;   // bytes of LDS the attention core uses for its two [rows][S + 1] float arrays

// query rows per block: as many as fit CORE_LDS next to all S keys (at most 64)
inline int core_rows(int T, int S) {
    int tc = CORE_LDS / (2 * (S + 1) * (int)sizeof(float));
    if (tc > 64) tc = 64;
    if (tc > T) tc = T;
    return tc < 1 ? 1 : tc;
}

struct CoreArgs {
    const void* q;        // [B*T,E] projected
    const void* k;        // [B*S,E]
    const void* v;        // [B*S,E]
    const float* attn_mask;
    int64_t mask_stride;
    const uint8_t* kpm;
    const float* drop_u;
    float drop_p;
    void* o;              // [B*T,E]
    float* probs;         // [B,H,T,S]
    float* attn_w;        // [B,T,S]
    // backward
    const void* dout;     // [B*T,E] gradient of o
    const float* d_attn_w;
    void* dq;
    void* dk;
    void* dv;
    float* dk32;          // [B*S,E] float32 accumulators over the query chunks (only when T > rows per chunk)
    float* dv32;
    int T, S, E, H;
    int tc;               // query rows per chunk
    float scale;
};

// Forward: one block per (sample, chunk of tc query rows).  Per head: scores by wave-level dots against all S keys,
// softmax + dropout per row, weighted sum of V.  sc / wb: [tc][S + 1] floats in dynamic LDS.
template <typename T>
__global__ __launch_bounds__(256) void mha_core_fwd_kernel(CoreArgs p) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    extern __shared__ __attribute__((aligned(16))) char core_smem[];
    const int Tn = p.T, S = p.S, E = p.E, H = p.H, hd = E / H, SP = S + 1;
    float* sc = reinterpret_cast<float*>(core_smem);
    float* wb = sc + p.tc * SP;
    const int64_t b = blockIdx.x;
    const int t0 = blockIdx.y * p.tc;
    const int nt = (Tn - t0) < p.tc ? (Tn - t0) : p.tc;
    const int lane = lane_id(), w = wave_id();
    const elem* q = reinterpret_cast<const elem*>(p.q) + (b * Tn + t0) * (int64_t)E;
    const elem* k = reinterpret_cast<const elem*>(p.k) + b * S * (int64_t)E;
    const elem* v = reinterpret_cast<const elem*>(p.v) + b * S * (int64_t)E;
    elem* o = reinterpret_cast<elem*>(p.o) + (b * Tn + t0) * (int64_t)E;
    for (int i = threadIdx.x; i < nt * S; i += 256) wb[(i / S) * SP + i % S] = 0.f;
    const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    for (int h = 0; h < H; ++h) {
        __syncthreads();
        for (int pair = w; pair < nt * S; pair += 4) {
            const int t = pair / S, s = pair % S;
            float a = 0.f;
            for (int e = lane; e < hd; e += 64)
                a += X::to_f32(q[(int64_t)t * E + h * hd + e]) * X::to_f32(k[(int64_t)s * E + h * hd + e]);
            a = reduce_wave(a);
            if (lane == 0) {
                a *= p.scale;
                if (p.attn_mask) a += p.attn_mask[(b * H + h) * p.mask_stride + (int64_t)(t0 + t) * S + s];
                if (p.kpm && p.kpm[b * S + s]) a = -INFINITY;
                sc[t * SP + s] = a;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nt; t += 256) {
            float mx = -INFINITY;
            for (int s = 0; s < S; ++s) mx = fmaxf(mx, sc[t * SP + s]);
            float sum = 0.f;
            for (int s = 0; s < S; ++s) { const float ex = expf(sc[t * SP + s] - mx); sc[t * SP + s] = ex; sum += ex; }
            for (int s = 0; s < S; ++s) {
                float pv = sc[t * SP + s] / sum;
                const int64_t idx = ((b * H + h) * Tn + t0 + t) * S + s;
                p.probs[idx] = pv;
                if (p.drop_p > 0.f) pv = p.drop_u[idx] >= p.drop_p ? pv * keep_scale : 0.f;
                sc[t * SP + s] = pv;
                wb[t * SP + s] += pv;
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nt * hd; idx += 256) {
            const int t = idx / hd, e = idx % hd;
            float a = 0.f;
            for (int s = 0; s < S; ++s) a += sc[t * SP + s] * X::to_f32(v[(int64_t)s * E + h * hd + e]);
            o[(int64_t)t * E + h * hd + e] = X::from_f32(a);
        }
    }
    __syncthreads();
    const float invH = 1.0f / (float)H;
    for (int i = threadIdx.x; i < nt * S; i += 256)
        p.attn_w[(b * Tn + t0) * S + i] = wb[(i / S) * SP + i % S] * invH;
}

// Backward: one block per sample, looping over the query chunks (dq rows are complete per chunk; dk / dv are sums over
// all query rows: element (s, e) is owned by one thread, which carries it over the chunks -- in float32 scratch when there
// is more than one chunk).
template <typename T>
__global__ __launch_bounds__(256) void mha_core_bwd_kernel(CoreArgs p) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    extern __shared__ __attribute__((aligned(16))) char core_smem[];
    const int Tn = p.T, S = p.S, E = p.E, H = p.H, hd = E / H, SP = S + 1;
    float* pp = reinterpret_cast<float*>(core_smem);     // dropped-out weights p'
    float* ds = pp + p.tc * SP;
    const int64_t b = blockIdx.x;
    const int lane = lane_id(), w = wave_id();
    const elem* k = reinterpret_cast<const elem*>(p.k) + b * S * (int64_t)E;
    const elem* v = reinterpret_cast<const elem*>(p.v) + b * S * (int64_t)E;
    elem* dk = reinterpret_cast<elem*>(p.dk) + b * S * (int64_t)E;
    elem* dv = reinterpret_cast<elem*>(p.dv) + b * S * (int64_t)E;
    float* dk32 = p.dk32 ? p.dk32 + b * S * (int64_t)E : nullptr;
    float* dv32 = p.dv32 ? p.dv32 + b * S * (int64_t)E : nullptr;
    const float keep_scale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const float invH = 1.0f / (float)H;
    const int nchunk = (Tn + p.tc - 1) / p.tc;
    for (int h = 0; h < H; ++h) {
        for (int ci = 0; ci < nchunk; ++ci) {
            const int t0 = ci * p.tc;
            const int nt = (Tn - t0) < p.tc ? (Tn - t0) : p.tc;
            const elem* q = reinterpret_cast<const elem*>(p.q) + (b * Tn + t0) * (int64_t)E;
            const elem* dout = reinterpret_cast<const elem*>(p.dout) + (b * Tn + t0) * (int64_t)E;
            elem* dq = reinterpret_cast<elem*>(p.dq) + (b * Tn + t0) * (int64_t)E;
            __syncthreads();
            // dp'[t][s] = dout_h[t] . V_h[s] + d_attn_w[t][s] / H
            for (int pair = w; pair < nt * S; pair += 4) {
                const int t = pair / S, s = pair % S;
                float a = 0.f;
                for (int e = lane; e < hd; e += 64)
                    a += X::to_f32(dout[(int64_t)t * E + h * hd + e]) * X::to_f32(v[(int64_t)s * E + h * hd + e]);
                a = reduce_wave(a);
                if (lane == 0) {
                    if (p.d_attn_w) a += p.d_attn_w[(b * Tn + t0 + t) * S + s] * invH;
                    ds[t * SP + s] = a;
                }
            }
            __syncthreads();
            for (int t = threadIdx.x; t < nt; t += 256) {
                float dot = 0.f;
                for (int s = 0; s < S; ++s) {
                    const int64_t idx = ((b * H + h) * Tn + t0 + t) * S + s;
                    const float pr = p.probs[idx];
                    float keep = 1.0f;
                    if (p.drop_p > 0.f) keep = p.drop_u[idx] >= p.drop_p ? keep_scale : 0.f;
                    const float dpr = ds[t * SP + s] * keep;        // gradient on the softmax output p
                    pp[t * SP + s] = pr * keep;                     // p' (what multiplied V)
                    ds[t * SP + s] = dpr;
                    dot += pr * dpr;
                }
                for (int s = 0; s < S; ++s) {
                    const int64_t idx = ((b * H + h) * Tn + t0 + t) * S + s;
                    ds[t * SP + s] = p.probs[idx] * (ds[t * SP + s] - dot) * p.scale;   // gradient on the scaled scores
                }
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < nt * hd; idx += 256) {
                const int t = idx / hd, e = idx % hd;
                float a = 0.f;
                for (int s = 0; s < S; ++s) a += ds[t * SP + s] * X::to_f32(k[(int64_t)s * E + h * hd + e]);
                dq[(int64_t)t * E + h * hd + e] = X::from_f32(a);
            }
            for (int idx = threadIdx.x; idx < S * hd; idx += 256) {
                const int s = idx / hd, e = idx % hd;
                float a = 0.f, c = 0.f;
                for (int t = 0; t < nt; ++t) {
                    a += ds[t * SP + s] * X::to_f32(q[(int64_t)t * E + h * hd + e]);
                    c += pp[t * SP + s] * X::to_f32(dout[(int64_t)t * E + h * hd + e]);
                }
                const int64_t o = (int64_t)s * E + h * hd + e;
                if (nchunk > 1) {
                    if (ci > 0) { a += dk32[o]; c += dv32[o]; }
                    if (ci + 1 < nchunk) { dk32[o] = a; dv32[o] = c; }
                }
                if (ci + 1 == nchunk) { dk[o] = X::from_f32(a); dv[o] = X::from_f32(c); }
            }
        }
    }
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline int esize(int dtype) { return dtype == AECF_BF16 ? 2 : 4; }
inline int launch_status() { return hipGetLastError() == hipSuccess ? AECF_OK : AECF_ERR_LAUNCH; }

void nt(int dtype, const void* a, int64_t rows, const void* w, const void* bias, void* c, int E, int H, hipStream_t s) {
    GemmNtArgs g;
    g.a = a; g.w = w; g.bias = bias; g.c = c; g.probs = nullptr; g.R = rows; g.N = E; g.K = E; g.lda = E;
    g.M = 1; g.H = H; g.hd = E / H; g.pooled = 0; g.out_f32 = 0; g.v_out = nullptr;
    launch_gemm_nt(dtype, g, s);
}

struct MhaWs {
    size_t wt[4], dob, dq, dk, dv, dk32, dv32, slab[4], cs[4], total;
    int splits[2];
    int64_t rps[2];
};
// splits for a batch-reduction GEMM over `rows` rows
void split_rows(int64_t rows, int E, int& S, int64_t& rps) {
    const int tiles = ((E + 127) / 128) * ((E + 127) / 128);
    S = (512 + tiles - 1) / tiles;
    const int64_t max_s = (rows + 255) / 256;
    if (S > max_s) S = (int)max_s;
    if (S < 1) S = 1;
    rps = (rows + S - 1) / S;
    rps = (rps + 31) / 32 * 32;
    S = (int)((rows + rps - 1) / rps);
}
MhaWs mha_layout(const aecf_mha_desc* d) {
    MhaWs w;
    const size_t E = d->embed_dim, es = esize(d->dtype);
    const size_t RT = (size_t)d->batch * d->tgt_len, RS = (size_t)d->batch * d->src_len;
    split_rows((int64_t)RT, (int)E, w.splits[0], w.rps[0]);     // reductions over B*T rows (q, out)
    split_rows((int64_t)RS, (int)E, w.splits[1], w.rps[1]);     // reductions over B*S rows (k, v)
    size_t off = 0;
    for (int i = 0; i < 4; ++i) { w.wt[i] = off; off = align_up(off + E * E * es); }
    w.dob = off; off = align_up(off + RT * E * es);
    w.dq = off;  off = align_up(off + RT * E * es);
    w.dk = off;  off = align_up(off + RS * E * es);
    w.dv = off;  off = align_up(off + RS * E * es);
    const bool chunked = core_rows(d->tgt_len, d->src_len) < d->tgt_len;     // float32 carries of dk / dv over the query chunks
    w.dk32 = off; off = align_up(off + (chunked ? RS * E * 4 : 0));
    w.dv32 = off; off = align_up(off + (chunked ? RS * E * 4 : 0));
    const int sp[4] = {w.splits[0], w.splits[1], w.splits[1], w.splits[0]};     // q, k, v, out
    for (int i = 0; i < 4; ++i) { w.slab[i] = off; off = align_up(off + (size_t)sp[i] * E * E * 4); }
    for (int i = 0; i < 4; ++i) { w.cs[i] = off; off = align_up(off + (size_t)sp[i] * E * 4); }
    w.total = off;
    return w;
}

void tn(int dtype, const void* lhs, const void* rhs, int64_t rows, float* slab, float* cs, int E, int H, int splits,
        int64_t rps, hipStream_t s) {
    GemmTnArgs t;
    t.lhs = lhs; t.rhs = rhs; t.probs = nullptr; t.dsbuf = nullptr; t.out = slab; t.colsum = cs; t.u = nullptr;
    t.B = rows; t.M = 1; t.E = E; t.H = H; t.hd = E / H; t.Ej = 0; t.splits = splits; t.rows_per_split = rps;
    t.u_splits = 0; t.u_rows_per_split = 0; t.pooled = 0;
    launch_gemm_tn(dtype, t, s);
}

}  // namespace

}  // namespace aecf

using namespace aecf;

extern "C" {

int aecf_mha_check(const aecf_mha_desc* d) {
    if (!d) return AECF_ERR_NULL_POINTER;
    if (d->batch <= 0 || d->tgt_len <= 0 || d->src_len <= 0 || d->embed_dim <= 0 || d->num_heads <= 0)
        return AECF_ERR_BAD_DIMS;
    if (d->embed_dim % d->num_heads != 0) return AECF_ERR_BAD_DIMS;
    if (d->dtype != AECF_BF16 && d->dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (d->tgt_len > LMAX || d->src_len > LMAX) return AECF_ERR_UNSUPPORTED;
    // any head count dividing E; E itself a whole number of 128-byte K slices of the GEMM tiles: a multiple of 32 (f32) /
    // 64 (bf16) -- E = 32, 96, ... in float32
    if (d->embed_dim % (d->dtype == AECF_BF16 ? 64 : 32) != 0 || d->embed_dim > 1024) return AECF_ERR_UNSUPPORTED;
    if (!(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return AECF_ERR_BAD_DIMS;
    return AECF_OK;
}

size_t aecf_mha_bwd_workspace_bytes(const aecf_mha_desc* d) {
    if (aecf_mha_check(d) != AECF_OK) return 0;
    return mha_layout(d).total;
}

int aecf_mha_forward(const aecf_mha_desc* d, const aecf_mha_fwd_args* a, void* stream) {
    int st = aecf_mha_check(d);
    if (st != AECF_OK) return st;
    if (!a || !a->query || !a->key || !a->value || !a->w_in || !a->w_out || !a->y || !a->attn_w || !a->saved_q ||
        !a->saved_k || !a->saved_v || !a->saved_o || !a->saved_probs)
        return AECF_ERR_NULL_POINTER;
    if (d->dropout_p > 0.f && !a->dropout_uniforms) return AECF_ERR_NULL_POINTER;
    hipStream_t s = (hipStream_t)stream;
    const int E = d->embed_dim, H = d->num_heads, es = esize(d->dtype);
    const int64_t RT = d->batch * d->tgt_len, RS = d->batch * d->src_len;
    const char* w = (const char*)a->w_in;
    const char* bi = (const char*)a->b_in;
    nt(d->dtype, a->query, RT, w, bi, a->saved_q, E, H, s);
    nt(d->dtype, a->key, RS, w + (size_t)E * E * es, bi ? bi + (size_t)E * es : nullptr, a->saved_k, E, H, s);
    nt(d->dtype, a->value, RS, w + (size_t)2 * E * E * es, bi ? bi + (size_t)2 * E * es : nullptr, a->saved_v, E, H, s);
    CoreArgs c = {};
    c.q = a->saved_q; c.k = a->saved_k; c.v = a->saved_v; c.attn_mask = a->attn_mask; c.mask_stride = a->attn_mask_stride;
    c.kpm = a->key_padding_mask; c.drop_u = a->dropout_uniforms; c.drop_p = d->dropout_p; c.o = a->saved_o;
    c.probs = a->saved_probs; c.attn_w = a->attn_w; c.T = d->tgt_len; c.S = d->src_len; c.E = E; c.H = H;
    c.scale = sqrtf(1.0f / (float)(E / H));
    c.tc = core_rows(d->tgt_len, d->src_len);
    {
        const size_t smem = (size_t)2 * c.tc * (d->src_len + 1) * sizeof(float);
        const dim3 grid((unsigned)d->batch, (unsigned)((d->tgt_len + c.tc - 1) / c.tc));
        if (d->dtype == AECF_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mha_core_fwd_kernel<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS + 4096);
            mha_core_fwd_kernel<BF16><<<grid, dim3(256), smem, s>>>(c);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mha_core_fwd_kernel<F32>), hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS + 4096);
            mha_core_fwd_kernel<F32><<<grid, dim3(256), smem, s>>>(c);
        }
    }
    nt(d->dtype, a->saved_o, RT, a->w_out, a->b_out, a->y, E, H, s);
    return launch_status();
}

int aecf_mha_backward(const aecf_mha_desc* d, const aecf_mha_bwd_args* a, void* stream) {
    int st = aecf_mha_check(d);
    if (st != AECF_OK) return st;
    if (!a || !a->query || !a->key || !a->value || !a->w_in || !a->w_out || !a->dy || !a->saved_q || !a->saved_k ||
        !a->saved_v || !a->saved_o || !a->saved_probs || !a->dquery || !a->dkey || !a->dvalue || !a->dw_in ||
        !a->db_in || !a->dw_out || !a->db_out || !a->workspace)
        return AECF_ERR_NULL_POINTER;
    if (d->dropout_p > 0.f && !a->dropout_uniforms) return AECF_ERR_NULL_POINTER;
    const MhaWs L = mha_layout(d);
    if (a->workspace_bytes < L.total) return AECF_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)a->workspace;
    const int E = d->embed_dim, H = d->num_heads, es = esize(d->dtype);
    const int64_t RT = d->batch * d->tgt_len, RS = d->batch * d->src_len;
    const char* w = (const char*)a->w_in;
    // transposed weights: the input gradients are NT GEMMs against W^T
    void* wt[4];
    for (int i = 0; i < 4; ++i) wt[i] = ws + L.wt[i];
    for (int i = 0; i < 3; ++i) launch_transpose(d->dtype, w + (size_t)i * E * E * es, wt[i], E, s);
    launch_transpose(d->dtype, a->w_out, wt[3], E, s);
    void* dob = ws + L.dob;
    void* dqb = ws + L.dq;
    void* dkb = ws + L.dk;
    void* dvb = ws + L.dv;
    nt(d->dtype, a->dy, RT, wt[3], nullptr, dob, E, H, s);                       // gradient of the head outputs
    CoreArgs c = {};
    c.q = a->saved_q; c.k = a->saved_k; c.v = a->saved_v; c.drop_u = a->dropout_uniforms; c.drop_p = d->dropout_p;
    c.probs = const_cast<float*>(a->saved_probs); c.dout = dob; c.d_attn_w = a->d_attn_w; c.dq = dqb; c.dk = dkb;
    c.dv = dvb; c.T = d->tgt_len; c.S = d->src_len; c.E = E; c.H = H; c.scale = sqrtf(1.0f / (float)(E / H));
    c.tc = core_rows(d->tgt_len, d->src_len);
    if (c.tc < d->tgt_len) { c.dk32 = (float*)(ws + L.dk32); c.dv32 = (float*)(ws + L.dv32); }
    {
        const size_t smem = (size_t)2 * c.tc * (d->src_len + 1) * sizeof(float);
        if (d->dtype == AECF_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mha_core_bwd_kernel<BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS + 4096);
            mha_core_bwd_kernel<BF16><<<dim3((unsigned)d->batch), dim3(256), smem, s>>>(c);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mha_core_bwd_kernel<F32>), hipFuncAttributeMaxDynamicSharedMemorySize, CORE_LDS + 4096);
            mha_core_bwd_kernel<F32><<<dim3((unsigned)d->batch), dim3(256), smem, s>>>(c);
        }
    }
    nt(d->dtype, dqb, RT, wt[0], nullptr, a->dquery, E, H, s);
    nt(d->dtype, dkb, RS, wt[1], nullptr, a->dkey, E, H, s);
    nt(d->dtype, dvb, RS, wt[2], nullptr, a->dvalue, E, H, s);
    // weight / bias gradients: batch reductions into float32 slabs, then one fixed-order reduction each
    float* slab[4];
    float* cs[4];
    for (int i = 0; i < 4; ++i) { slab[i] = (float*)(ws + L.slab[i]); cs[i] = (float*)(ws + L.cs[i]); }
    tn(d->dtype, dqb, a->query, RT, slab[0], cs[0], E, H, L.splits[0], L.rps[0], s);
    tn(d->dtype, dkb, a->key, RS, slab[1], cs[1], E, H, L.splits[1], L.rps[1], s);
    tn(d->dtype, dvb, a->value, RS, slab[2], cs[2], E, H, L.splits[1], L.rps[1], s);
    tn(d->dtype, a->dy, a->saved_o, RT, slab[3], cs[3], E, H, L.splits[0], L.rps[0], s);
    const int sp[4] = {L.splits[0], L.splits[1], L.splits[1], L.splits[0]};
    ReduceSegs r1 = {}, r2 = {};
    for (int i = 0; i < 3; ++i) {
        r1.src[i] = slab[i]; r1.dst[i] = a->dw_in + (size_t)i * E * E; r1.n[i] = (int64_t)E * E; r1.splits[i] = sp[i];
        r2.src[i] = cs[i];   r2.dst[i] = a->db_in + (size_t)i * E;     r2.n[i] = E;              r2.splits[i] = sp[i];
    }
    r1.src[3] = slab[3]; r1.dst[3] = a->dw_out; r1.n[3] = (int64_t)E * E; r1.splits[3] = sp[3];
    r2.src[3] = cs[3];   r2.dst[3] = a->db_out; r2.n[3] = E;              r2.splits[3] = sp[3];
    launch_reduce_segments(r1, s);
    launch_reduce_segments(r2, s);
    return launch_status();
}

}  // extern "C"
