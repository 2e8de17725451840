// Device-side building blocks shared by every kernel of libaecf_hip (gfx950 / CDNA4 only).
//
// Conventions used throughout:
//   * one wavefront = 64 lanes; r16 = lane & 15, lg = lane >> 4.
//   * MFMA tiles are 16x16 (v_mfma_f32_16x16x32_bf16 for bf16, v_mfma_f32_16x16x4_f32 for f32).
//     A "fragment" is the 16 bytes one lane contributes to one K-step:
//        bf16: 8 consecutive k (KSTEP = 32 per MFMA),  A[row r16][k0 + 8*lg + j]
//        f32 : 4 consecutive k (KSTEP = 16 = 4 MFMAs), A[row r16][k0 + 4*lg + t] feeds MFMA t
//     (the K order inside a step is a free choice as long as A and B agree).
//   * accumulator (C/D) layout of a 16x16 tile: col = r16, row = 4*lg + reg   (reg = 0..3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct BF16 {};
struct F32 {};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned int b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ unsigned int f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;            // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return (unsigned int)__builtin_bit_cast(unsigned short, h);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    bf16x2 v = __builtin_convertvector(f32x2{lo, hi}, bf16x2);      // ONE v_cvt_pk_bf16_f32 (both halves, RNE)
    return __builtin_bit_cast(unsigned int, v);
}

template <typename T> struct Tr;

template <> struct Tr<BF16> {
    typedef unsigned short elem;
    typedef u32x4 frag;                 // 8 bf16
    static constexpr int EPL = 8;       // elements per lane per K-step
    static constexpr int KSTEP = 32;
    static constexpr int BYTES = 2;
    static __device__ __forceinline__ frag zero() { return frag{0u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ frag load(const elem* p) { return *reinterpret_cast<const frag*>(p); }
    static __device__ __forceinline__ void unpack(frag f, float* o) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(f[i] << 16);
            o[2 * i + 1] = __uint_as_float(f[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ frag pack(const float* v) {
        frag f;
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        return f;
    }
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
    }
    // 4 consecutive elements (accumulator-layout rows)
    static __device__ __forceinline__ void load4(const elem* p, float* o) {
        u32x2 v = *reinterpret_cast<const u32x2*>(p);
        o[0] = __uint_as_float(v[0] << 16);
        o[1] = __uint_as_float(v[0] & 0xffff0000u);
        o[2] = __uint_as_float(v[1] << 16);
        o[3] = __uint_as_float(v[1] & 0xffff0000u);
    }
    static __device__ __forceinline__ void store4(elem* p, const float* v) {
        u32x2 o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(p) = o;
    }
    static __device__ __forceinline__ float to_f32(elem e) { return __uint_as_float(((unsigned int)e) << 16); }
    static __device__ __forceinline__ elem from_f32(float f) { return (elem)f32_to_bf16_bits(f); }
};

template <> struct Tr<F32> {
    typedef float elem;
    typedef f32x4 frag;                 // 4 f32
    static constexpr int EPL = 4;
    static constexpr int KSTEP = 16;
    static constexpr int BYTES = 4;
    static __device__ __forceinline__ frag zero() { return frag{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ frag load(const elem* p) { return *reinterpret_cast<const frag*>(p); }
    static __device__ __forceinline__ void unpack(frag f, float* o) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f[i];
    }
    static __device__ __forceinline__ frag pack(const float* v) { return frag{v[0], v[1], v[2], v[3]}; }
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
#pragma unroll
        for (int t = 0; t < 4; ++t) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ void load4(const elem* p, float* o) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    static __device__ __forceinline__ void store4(elem* p, const float* v) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
    static __device__ __forceinline__ float to_f32(elem e) { return e; }
    static __device__ __forceinline__ elem from_f32(float f) { return f; }
};

// parameter-gradient element store: float32, or bf16 (one rounding of the float32 value)
__device__ __forceinline__ void store_grad(void* base, int64_t i, float v, int bf16) {
    if (bf16) reinterpret_cast<unsigned short*>(base)[i] = (unsigned short)f32_to_bf16_bits(v);
    else reinterpret_cast<float*>(base)[i] = v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// sum over the 4 lane groups (lanes l, l^16, l^32, l^48): every lane ends with the total
__device__ __forceinline__ float reduce_lg(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// sum over the 16 lanes of one lane group
__device__ __forceinline__ float reduce_r16(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ float reduce_wave(float v) { return reduce_lg(reduce_r16(v)); }

// ---------------------------------------------------------------------------------------------
// dq'[j] = scale * W_k[j,:] . u[h(j)]  (the query-side gradient every remaining piece of the backward's tail hangs on:
// dW_q = dq' (x) q, db_q = dq', dquery = W_q^T dq').  E dots of length E against the REDUCED u: small enough to ride as a side
// job in the weight-gradient launch that runs between the kernel that reduces u and the finalize launch (rows dealt over the
// launch's blocks, one wave per row), so that the finalize launch has no dependent chain inside it.
struct DqpJob {
    const void* w_k = nullptr;    // [E,E] dtype (rows j); null = off
    const float* u = nullptr;     // [H,E] reduced
    float* dqp = nullptr;         // [E]
    float scale = 0.f;
    int E = 0, hd = 1;
    // optional (stand-alone launch, a few slabs): u is still in u_nslab partial slabs [u_nslab][H, E]; every row adds them up for
    // its head on the fly and the first row of a head writes the sums to u_out [H, E] (what the finalize launch reads)
    const float* u_slab = nullptr;
    float* u_out = nullptr;
    int u_nslab = 0;
};

template <typename T>
__device__ __forceinline__ void dqp_rows(const DqpJob& q, int block, int nblocks) {
    using X = Tr<T>;
    const int per = (q.E + nblocks - 1) / nblocks;
    const int j0 = block * per, j1 = (j0 + per) < q.E ? (j0 + per) : q.E;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (int)blockDim.x >> 6;
    for (int j = j0 + w; j < j1; j += nw) {                        // (wave-uniform bounds: whole waves take the butterfly)
        const int64_t urow = (int64_t)(j / q.hd) * q.E;
        const float* u = q.u + urow;
        const typename X::elem* wr = reinterpret_cast<const typename X::elem*>(q.w_k) + (int64_t)j * q.E;
        float a = 0.f;
        for (int c = lane * 8; c < q.E; c += 512) {
            float wv[8];
            X::load4(wr + c, wv);
            X::load4(wr + c + 4, wv + 4);
            f32x4 u0, u1;
            if (q.u_slab) {
                u0 = u1 = f32x4{0.f, 0.f, 0.f, 0.f};
                const int64_t he = (int64_t)(q.E / q.hd) * q.E;
                for (int sl = 0; sl < q.u_nslab; ++sl) {           // (fixed slab order: deterministic)
                    const float* us = q.u_slab + sl * he + urow + c;
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(us), p1 = *reinterpret_cast<const f32x4*>(us + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { u0[e] += p0[e]; u1[e] += p1[e]; }
                }
                if (j % q.hd == 0) {
                    *reinterpret_cast<f32x4*>(q.u_out + urow + c) = u0;
                    *reinterpret_cast<f32x4*>(q.u_out + urow + c + 4) = u1;
                }
            } else {
                u0 = *reinterpret_cast<const f32x4*>(u + c);
                u1 = *reinterpret_cast<const f32x4*>(u + c + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a = fmaf(wv[e], u0[e], a);
#pragma unroll
            for (int e = 0; e < 4; ++e) a = fmaf(wv[4 + e], u1[e], a);
        }
        a = reduce_wave(a);
        if (lane == 0) q.dqp[j] = a * q.scale;
    }
}

// ---------------------------------------------------------------------------------------------
// The Bernoulli draw of ref aecf/AECFLayer.py:204 without a uniforms tensor: the statistics kernel evaluates, per weight
// element, the SAME counter-based generator call that `torch.rand(n, device=...)` makes for that element, so a step that
// lets the kernel draw (aecf_pool_fwd_args.flags & AECF_DRAW_UNIFORMS) and a step that passes `uniforms = torch.rand(...)`
// from the same generator state see the same masks bit for bit, and the generator advances by the same offset.
//   torch (ATen/native/cuda/DistributionTemplates.h: distribution_elementwise_grid_stride_kernel, unroll 4) launches
//   T = 256 * grid threads; thread idx seeds Philox4x32-10 with (seed, subsequence = idx, offset) and per grid-stride
//   iteration draws ONE 4-vector; component ii of iteration `it` goes to element  li = idx + T * ii + 4 T * it.
//   rocRAND's engine: counter = {offset / 4 + it (64-bit: x, y), subsequence (64-bit: z, w)}, key = seed; value =
//   2^-32 + v * 2^-32  in (0, 1], and torch maps 1.0 to 0.0 ("reverse the bounds").
// Philox4x32-10 itself is Salmon et al., SC'11 (Random123); known-answer vectors in tests/test_host_cpu.py.
struct PhiloxDraw {
    unsigned long long seed = 0, offset = 0;      // the generator's (seed, philox offset) BEFORE this draw
    unsigned int threads = 0;                     // T of the torch launch this draw replaces (0 = drawing off)
    unsigned long long elem0 = 0;                 // element of that launch this call's first weight element is (a rank's rows of
                                                  // ONE global draw: first global row * M; 0 = the call is the whole draw)
};

__device__ __host__ __forceinline__ void philox4x32_10(unsigned int c[4], unsigned int k0, unsigned int k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (unsigned int)p1;
        c[3] = (unsigned int)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// the float32 uniform torch.rand puts at linear element li
__device__ __host__ __forceinline__ float philox_uniform_at(const PhiloxDraw& ph, long long local) {
    const unsigned long long li = (unsigned long long)local + ph.elem0;
    const unsigned long long T = ph.threads, per = 4ull * T;
    const unsigned long long it = li / per, rem = li - it * per;
    const unsigned int ii = (unsigned int)(rem / T);
    const unsigned long long idx = rem - (unsigned long long)ii * T;
    const unsigned long long ctr = ph.offset / 4ull + it;
    unsigned int c[4] = {(unsigned int)ctr, (unsigned int)(ctr >> 32), (unsigned int)idx, (unsigned int)(idx >> 32)};
    philox4x32_10(c, (unsigned int)ph.seed, (unsigned int)(ph.seed >> 32));
    const float u = 2.3283064e-10f + (float)c[ii] * 2.3283064e-10f;
    return u == 1.0f ? 0.0f : u;
}

// ---------------------------------------------------------------------------------------------
// CurriculumMasking row arithmetic (ref aecf/AECFLayer.py:130-283), one row of length L in
// registers.  Shared by the fused gate kernel and the stand-alone mask kernel so both produce
// bit-identical results for the same (weights, uniforms).
// ---------------------------------------------------------------------------------------------
struct MaskCfg {
    int mode;             // 0 none, 1 train, 2 eval
    int min_active;
    float base_mask_prob;
    float entropy_target;
    float eps;
    float log_L;          // (float)log((double)L), computed on the host like the reference's math.log
    float inv_L;          // (float)(1.0 / L)
};

// x*log(x) with xlogy semantics: 0 at x==0, NaN propagates (torch.xlogy, ref :125)
__device__ __forceinline__ float xlogx(float w) { return (w == 0.f) ? 0.f : w * logf(w); }

// BITS: the word that holds one keep bit per key (unsigned int up to 32 keys, unsigned long long up to 64)
template <int LMAX, typename BITS = unsigned int>
__device__ __forceinline__ void curriculum_row(const MaskCfg& c, int L, float* w /*in: weights, out: normalised*/,
                                               const float* u, float* masked, float& entropy, float& mask_rate,
                                               BITS& mask_bits) {
    const float logL = c.log_L;
    if (c.mode == 2) {  // eval (ref :150-156): weights unchanged, entropy of the raw weights
        float h = 0.f;
#pragma unroll
        for (int i = 0; i < LMAX; ++i)
            if (i < L) { h -= xlogx(w[i]); masked[i] = w[i]; }
        entropy = fminf(fmaxf(h, 0.f), logL);
        if (h != h) entropy = h;
        mask_rate = 0.f;
        mask_bits = ~(BITS)0;
        return;
    }
    if (L <= 1) {  // ref :160-167
        masked[0] = w[0];
        entropy = 0.f;
        mask_rate = 0.f;
        mask_bits = (BITS)1;
        return;
    }
    // ref :170-184  (non-finite entries -> 0; rows summing below eps -> uniform; else w / sum)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LMAX; ++i)
        if (i < L) {
            if (!isfinite(w[i])) w[i] = 0.f;
            s += w[i];
        }
    const bool needs_norm = s < c.eps;
#pragma unroll
    for (int i = 0; i < LMAX; ++i)
        if (i < L) w[i] = needs_norm ? c.inv_L : (w[i] / s);
    // ref :190-201
    float h = 0.f;
#pragma unroll
    for (int i = 0; i < LMAX; ++i)
        if (i < L) h -= xlogx(w[i]);
    h = fminf(fmaxf(h, 0.f), logL);
    entropy = h;
    float ne = fminf(fmaxf(h / logL, 0.f), 1.f);
    float keep = fminf(fmaxf(1.0f - c.base_mask_prob * ne, 0.f), 1.f);
    // ref :204  bernoulli(keep) == (u < keep) on the float32 uniform stream
    BITS bits = 0;
    int active = 0;
#pragma unroll
    for (int i = 0; i < LMAX; ++i)
        if (i < L && u[i] < keep) { bits |= ((BITS)1 << i); ++active; }
    // ref :207-260  min-active fix: rows with too few survivors keep exactly their top-k weights
    const int k = c.min_active < L ? c.min_active : L;
    if (active < k) {
        bits = 0;
        for (int t = 0; t < k; ++t) {   // k selections of the largest remaining, lowest index on ties
            int best = -1;
            float bv = 0.f;
#pragma unroll
            for (int i = 0; i < LMAX; ++i)
                if (i < L && !((bits >> i) & 1) && (best < 0 || w[i] > bv)) { best = i; bv = w[i]; }
            bits |= ((BITS)1 << best);
        }
        active = k;
    }
    // ref :263-272
    float ms = 0.f;
#pragma unroll
    for (int i = 0; i < LMAX; ++i)
        if (i < L) { masked[i] = ((bits >> i) & 1) ? w[i] : 0.f; ms += masked[i]; }
    const bool valid = ms > c.eps;
#pragma unroll
    for (int i = 0; i < LMAX; ++i)
        if (i < L) masked[i] = valid ? (masked[i] / ms) : w[i];
    mask_rate = 1.0f - (float)active / (float)L;   // ref :275
    mask_bits = bits;
}

#define AECF_DISPATCH_M(M, ...)                               \
    switch (M) {                                              \
        case 1: { constexpr int M_ = 1; __VA_ARGS__; break; } \
        case 2: { constexpr int M_ = 2; __VA_ARGS__; break; } \
        case 3: { constexpr int M_ = 3; __VA_ARGS__; break; } \
        case 4: { constexpr int M_ = 4; __VA_ARGS__; break; } \
        case 5: { constexpr int M_ = 5; __VA_ARGS__; break; } \
        case 6: { constexpr int M_ = 6; __VA_ARGS__; break; } \
        case 7: { constexpr int M_ = 7; __VA_ARGS__; break; } \
        case 8: { constexpr int M_ = 8; __VA_ARGS__; break; } \
        default: break;                                       \
    }
