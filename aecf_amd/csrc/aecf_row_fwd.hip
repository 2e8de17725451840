// Fused forward of the fusion pool, bf16, gfx950 -- ONE kernel from x to y (BASELINE.json north_star: scores, softmax,
// entropy-gated curriculum mask, value projection, pooling and out-projection in one launch):
//
//     s[b,h,m] = x[b,m] . A[h]                      scores against the folded key matrix (hi/lo bf16 split)
//     p        = softmax_m(s)  (key_padding_mask -> -inf)
//     V[b,m]   = W_v x[b,m] + b_v ;   o[b] = sum_m p[b,head(n),m] V[b,m]
//     y[b]     = W_o o[b] + b_o
//     wbar = mean_h p ; CurriculumMasking on wbar (ref aecf/AECFLayer.py:130-283)
//
// Row-stationary, weights streamed: a block owns a tile of TS = 16 ST samples and ALL E output columns, so the pooled
// heads o never leave the CU before the out-projection consumes them.  The two E x E weight matrices do not fit a CU's
// registers together with a full-row accumulator, so they are streamed from L2 (they are 1 MB, resident in every XCD's
// L2) through LDS in K-chunks of 32 by the DMA engine (global_load_lds_dwordx4, two slots: the copy of chunk c+1 flies
// behind the MFMAs of chunk c), together with the matching 64-byte column slice of the tile's x rows.  The product is
// formed transposed (weights = MFMA A operand), so a lane's accumulators are 4 consecutive output columns of ONE sample
// for every modality: the softmax weights are per-lane scalars and pooling is a per-lane FMA.
//   wave w of 8:  output columns [16 NW w, 16 NW (w+1)), all samples of the tile, all modalities
//                 accumulators NW x (ST M) tiles of 16x16 (E = 512, ST = 2, M = 3: 96 registers)
//   scores:       waves 0 .. ST M - 1 add one MFMA pair per chunk (A hi/lo rows = heads, LDS-resident)
// HBM traffic per sample: x read once (M E s bytes), y written once, plus what the backward asked to keep (o, V).
// LDS images have 64-byte rows (one MFMA K-step); 16-byte piece p of row r sits at piece p ^ ((r >> 2) & 2), which
// makes the ds_read_b128 fragment reads conflict-free for the hardware's 16-lane service groups (MI355X_MICROARCH.md,
// LDS table); the DMA destination is lane-linear, so the permutation is applied to the per-lane SOURCE address.
#include <stdlib.h>

#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

namespace {

__device__ __forceinline__ int sw64(int row) { return (row >> 2) & 2; }

// one wave-instruction of LDS-DMA: 16 rows x 64 bytes (1 KB).  Lane i fetches row i >> 2 (clamped to row_max), logical
// piece (i & 3) ^ sw64(row); the destination is the wave-uniform lds address + 16 i.  Issued through inline asm so that
// hipcc's waitcnt insertion does not serialise it with the LDS reads of the step it flies behind; the caller retires
// it with its own s_waitcnt vmcnt(0).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16x64(const char* __restrict__ base /*wave-uniform*/, unsigned int voff, char* lds) {
    const unsigned int dst = (unsigned)(size_t)(lds_void_t*)lds;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst)
                 : "memory", "m0");
}
#pragma clang diagnostic pop

// raw workgroup barrier with compiler fences either side (LDS traffic is ordered by the explicit waitcnts around it)
__device__ __forceinline__ void block_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct RowFwdArgs {
    const unsigned short* x;        // [B,M,E]
    const unsigned short* w_v;      // [E,E] rows n
    const unsigned short* b_v;      // [E] or null
    const unsigned short* w_o;      // [E,E]
    const unsigned short* b_o;      // [E] or null
    const unsigned short* a_hi;     // [HPAD,E] folded key matrix, bf16 hi / lo
    const unsigned short* a_lo;
    const uint8_t* kpm;             // [B,M] or null
    const float* uniforms;          // [B,M] or null
    PhiloxDraw ph;                  // drawn in the kernel when uniforms is null (aecf_common.h)
    unsigned short* y;              // [B,E]
    unsigned short* o;              // [B,E] saved heads or null
    unsigned short* v;              // [B,M,E] saved value projections or null
    float* probs;                   // [B,H,M]
    float* attn_w;                  // [B,M]
    float* masked_w;                // [B,M] or null
    float* entropy;                 // [B] or null
    float* mask_rate;               // [B] or null
    unsigned short* i_attn_w;       // info copies (bf16) or null
    unsigned short* i_masked_w;
    unsigned short* i_entropy;
    unsigned short* i_mask_rate;
    unsigned short* i_target;
    float target_value;
    int64_t B;
    int H, hd;
    MaskCfg mask;
};

template <int NW, int M_, int ST>
struct RowFwdCfg {
    static constexpr int E = 128 * NW, KT = E / 32, TS = 16 * ST, XT = ST * M_;
    static constexpr int WSLOT = E * 64;                      // one weight chunk: E rows x 64 B
    static constexpr int XSLOT = 8 * 1024;                    // x chunk: up to 128 rows x 64 B (one wave-instruction each)
    static constexpr int SLOT = WSLOT + XSLOT;
    static constexpr int OFF_A = 2 * SLOT;                    // A hi | A lo: [KT][16 rows][64 B] each
    static constexpr int OFF_O = OFF_A + 2 * KT * 1024;       // o tile: [KT][TS rows][64 B]
    static constexpr int OFF_SC = OFF_O + KT * TS * 64;       // scores [XT][16 heads][16 samples] f32
    static constexpr int OFF_PR = OFF_SC + XT * 1024;         // probs [TS][H][M] f32, H <= 8 (a wave's columns lie in one head)
    static constexpr int OFF_BIAS = OFF_PR + TS * 8 * M_ * 4; // b_v | b_o as f32
    static constexpr int TOTAL = OFF_BIAS + 2 * E * 4;
    static constexpr int YROW = 2 * E + 16;                   // y staging row pitch (bank spread for 8-byte writes)
    static_assert(TS * YROW <= SLOT, "y staging fits one ring slot");
    static_assert(XT <= 8, "one x wave-instruction per wave");
};

template <int NW, int M_, int ST>
__global__ __launch_bounds__(512, 2) void row_fwd_kernel(RowFwdArgs p, int ntile) {
    using C = RowFwdCfg<NW, M_, ST>;
    using X = Tr<BF16>;
    constexpr int E = C::E, KT = C::KT, TS = C::TS, XT = C::XT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());    // wave-uniform in an SGPR: scalar branches, m0 values
    const int H = p.H, hd = p.hd;
    const int64_t B = p.B;
    const int my_tiles = ((int)blockIdx.x < ntile) ? (ntile - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;

    // ---- per-lane constants
    const int dr = lane >> 2;                                   // DMA: row within the 16-row wave-instruction
    const unsigned int dpiece = (unsigned)(((lane & 3) ^ sw64(dr)) << 4);
    const int fsw = (lg ^ sw64(r16)) << 4;                      // fragment read: byte offset of this lane's piece
    const int ncol0 = 16 * NW * w;                              // this wave's first output column
    const int head = ncol0 / hd;                                // (16 NW divides hd)

    // ---- one-time LDS images: folded key matrix (hi/lo) by DMA, biases as float
    {
        for (int q = w; q < 2 * KT; q += 8) {                   // image q: chunk q % KT of (q < KT ? hi : lo)
            const unsigned short* src = q < KT ? p.a_hi : p.a_lo;
            const int c = q < KT ? q : q - KT;
            dma16x64(reinterpret_cast<const char*>(src), (unsigned)(dr * E * 2 + c * 64) + dpiece, smem + C::OFF_A + q * 1024);
        }
        float* bias = reinterpret_cast<float*>(smem + C::OFF_BIAS);
        for (int i = threadIdx.x; i < E; i += 512) {
            bias[i] = p.b_v ? X::to_f32(p.b_v[i]) : 0.f;
            bias[E + i] = p.b_o ? X::to_f32(p.b_o[i]) : 0.f;
        }
    }

    // chunk stream: (tile, phase 0 = W_v + x | phase 1 = W_o, chunk c) -> ring slot
    auto issue = [&](int tile, int phase, int c, int slot) {
        char* dst = smem + slot * C::SLOT;
        const char* wsrc = reinterpret_cast<const char*>(phase == 0 ? p.w_v : p.w_o);
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int row = (w * NW + q) * 16 + dr;
            dma16x64(wsrc, (unsigned)(row * E * 2 + c * 64) + dpiece, dst + (w * NW + q) * 1024);
        }
        if (phase == 0) {
            const int t = w < XT ? w : XT - 1;                  // x tile (st, m) of this wave (spare waves repeat the last)
            const int st = t / M_, m = t - st * M_;
            int64_t b = (int64_t)tile * TS + st * 16 + dr;
            b = b < B ? b : B - 1;
            const char* xsrc = reinterpret_cast<const char*>(p.x) + (b * M_ + m) * (int64_t)(E * 2);
            // (per-lane 64-bit row address folded into the scalar base is not possible: rows differ per lane; the
            //  offset from x stays below 2^32 only for B*M*E*2 < 4 GB, so the row base goes through a flat form)
            const unsigned int dst_lds = (unsigned)(size_t)(lds_void_t*)(dst + C::WSLOT + w * 1024);
            const char* src = xsrc + c * 64 + dpiece;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst_lds) : "memory", "m0");
#pragma clang diagnostic pop
        }
    };

    issue((int)blockIdx.x, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // bias image written (made visible by the first barrier)

    for (int ti = 0; ti < my_tiles; ++ti) {
        const int tile = (int)blockIdx.x + ti * (int)gridDim.x;
        const int64_t b0 = (int64_t)tile * TS;
        unsigned int kp[ST][M_];                                // key_padding_mask bytes of this lane's samples
#pragma unroll
        for (int st = 0; st < ST; ++st)
#pragma unroll
            for (int m = 0; m < M_; ++m) kp[st][m] = 0u;

        // ================= phase 0: V_m = W_v x_m for all modalities, scores =================
        f32x4 acc[NW][XT];
#pragma unroll
        for (int nt = 0; nt < NW; ++nt)
#pragma unroll
            for (int t = 0; t < XT; ++t) acc[nt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c = 0; c < KT; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            block_sync();
            if (c + 1 < KT) issue(tile, 0, c + 1, (c + 1) & 1);
            else {
                issue(tile, 1, 0, 0);
                if (p.kpm) {                                    // fetched behind the last chunk's MFMAs, used in the epilogue
#pragma unroll
                    for (int st = 0; st < ST; ++st)
#pragma unroll
                        for (int m = 0; m < M_; ++m) {
                            int64_t b = b0 + st * 16 + r16;
                            b = b < B ? b : B - 1;
                            kp[st][m] = (unsigned)p.kpm[b * M_ + m];
                        }
                }
            }
            const char* ws = smem + (c & 1) * C::SLOT;
            const char* xs = ws + C::WSLOT;
            u32x4 bf[XT];
#pragma unroll
            for (int t = 0; t < XT; ++t) bf[t] = *reinterpret_cast<const u32x4*>(xs + (t * 16 + r16) * 64 + fsw);
#pragma unroll
            for (int nt = 0; nt < NW; ++nt) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(ws + ((w * NW + nt) * 16 + r16) * 64 + fsw);
#pragma unroll
                for (int t = 0; t < XT; ++t) acc[nt][t] = X::mma(af, bf[t], acc[nt][t]);
            }
            if (w < XT) {                                       // wave-uniform: this wave's score tile
                const u32x4 xb = *reinterpret_cast<const u32x4*>(xs + (w * 16 + r16) * 64 + fsw);
                const u32x4 ah = *reinterpret_cast<const u32x4*>(smem + C::OFF_A + c * 1024 + r16 * 64 + fsw);
                const u32x4 al = *reinterpret_cast<const u32x4*>(smem + C::OFF_A + (KT + c) * 1024 + r16 * 64 + fsw);
                sacc = X::mma(ah, xb, sacc);
                sacc = X::mma(al, xb, sacc);
            }
        }

        // ---- epilogue 0: softmax weights, pooled heads o -> LDS (B operand of the out-projection), statistics
        float* sc = reinterpret_cast<float*>(smem + C::OFF_SC);
        float* pr = reinterpret_cast<float*>(smem + C::OFF_PR);
        if (w < XT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[(w * 16 + 4 * lg + r) * 16 + r16] = sacc[r];      // [tile][head][sample]
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        block_sync();
        float pm[ST][M_];
#pragma unroll
        for (int st = 0; st < ST; ++st) {
            float s[M_], mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                float a = sc[((st * M_ + m) * 16 + head) * 16 + r16];
                if (kp[st][m] != 0u) a = -INFINITY;             // torch functional.py:6554-6566
                s[m] = a;
                mx = fmaxf(mx, a);
            }
            float sum = 0.f;
#pragma unroll
            for (int m = 0; m < M_; ++m) { s[m] = expf(s[m] - mx); sum += s[m]; }
#pragma unroll
            for (int m = 0; m < M_; ++m) pm[st][m] = s[m] / sum;
            if (lg == 0 && ncol0 % hd == 0) {                   // first wave of the head, one lane per sample
#pragma unroll
                for (int m = 0; m < M_; ++m) pr[((st * 16 + r16) * H + head) * M_ + m] = pm[st][m];
            }
        }
        {
            const float* bias = reinterpret_cast<const float*>(smem + C::OFF_BIAS);
            char* ot = smem + C::OFF_O;
#pragma unroll
            for (int nt = 0; nt < NW; ++nt) {
                const int n = ncol0 + 16 * nt + 4 * lg;         // this lane's 4 consecutive columns
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                for (int st = 0; st < ST; ++st) {
                    float ov[4] = {0.f, 0.f, 0.f, 0.f};
                    const int64_t b = b0 + st * 16 + r16;
#pragma unroll
                    for (int m = 0; m < M_; ++m) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc[nt][st * M_ + m][r] + bv[r];
                            ov[r] = fmaf(pm[st][m], v[r], ov[r]);
                        }
                        if (p.v && b < B)
                            *reinterpret_cast<u32x2*>(p.v + (b * M_ + m) * E + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    }
                    // o tile image: chunk n / 32, row st*16 + r16, piece (n % 32) / 8, half (n % 8) / 4
                    const int row = st * 16 + r16;
                    const int off = (n >> 5) * (TS * 64) + row * 64 + (((((n & 31) >> 3)) ^ sw64(row)) << 4) + ((n & 4) << 1);
                    *reinterpret_cast<u32x2*>(ot + off) = u32x2{pack_bf16x2(ov[0], ov[1]), pack_bf16x2(ov[2], ov[3])};
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        block_sync();                           // o tile and probabilities complete

        // per-sample statistics: head mean, curriculum masking, info copies (one thread per sample)
        if (threadIdx.x < TS) {
            const int sidx = threadIdx.x;
            const int64_t bs = b0 + sidx;
            if (bs < B) {
                float wsel[M_];
#pragma unroll
                for (int m = 0; m < M_; ++m) wsel[m] = 0.f;
                for (int h = 0; h < H; ++h)
#pragma unroll
                    for (int m = 0; m < M_; ++m) wsel[m] += pr[(sidx * H + h) * M_ + m];
                const float invH = 1.0f / (float)H;
#pragma unroll
                for (int m = 0; m < M_; ++m) {
                    wsel[m] *= invH;
                    p.attn_w[bs * M_ + m] = wsel[m];
                    if (p.i_attn_w) p.i_attn_w[bs * M_ + m] = X::from_f32(wsel[m]);
                }
                if (p.mask.mode != 0) {
                    float wv[M_], u[M_], mk[M_];
#pragma unroll
                    for (int m = 0; m < M_; ++m) {
                        wv[m] = wsel[m];
                        u[m] = p.mask.mode != 1 ? 0.f : (p.uniforms ? p.uniforms[bs * M_ + m] : (p.ph.threads ? philox_uniform_at(p.ph, bs * M_ + m) : 0.f));
                    }
                    float ent, rate;
                    unsigned int bits;
                    curriculum_row<M_>(p.mask, M_, wv, u, mk, ent, rate, bits);
#pragma unroll
                    for (int m = 0; m < M_; ++m) {
                        if (p.masked_w) p.masked_w[bs * M_ + m] = mk[m];
                        if (p.i_masked_w) p.i_masked_w[bs * M_ + m] = X::from_f32(mk[m]);
                    }
                    if (p.entropy) p.entropy[bs] = ent;
                    if (p.mask_rate) p.mask_rate[bs] = rate;
                    if (p.i_entropy) p.i_entropy[bs] = X::from_f32(ent);
                    if (p.i_mask_rate) p.i_mask_rate[bs] = X::from_f32(rate);
                    if (p.i_target) p.i_target[bs] = X::from_f32(p.target_value);
                }
            }
        } else if (threadIdx.x >= 64) {
            // saved probabilities [b][h][m] (contiguous over the tile's samples) from the LDS copy, coalesced
            const int nvalid = (int)((B - b0) < TS ? (B - b0) : TS);
            for (int i = threadIdx.x - 64; i < nvalid * H * M_; i += 512 - 64) p.probs[b0 * H * M_ + i] = pr[i];
        }
        if (p.o) {                                              // saved heads: rows of the o tile, 16 bytes per thread
            const char* ot = smem + C::OFF_O;
            for (int i = threadIdx.x; i < TS * (E / 8); i += 512) {
                const int row = i / (E / 8), ch = i - row * (E / 8);          // 16-byte chunk ch of row
                const int off = (ch >> 2) * (TS * 64) + row * 64 + (((ch & 3) ^ sw64(row)) << 4);
                if (b0 + row < B) *reinterpret_cast<u32x4*>(p.o + (b0 + row) * E + ch * 8) = *reinterpret_cast<const u32x4*>(ot + off);
            }
        }

        // ================= phase 1: y = W_o o + b_o =================
        f32x4 yacc[NW][ST];
#pragma unroll
        for (int nt = 0; nt < NW; ++nt)
#pragma unroll
            for (int st = 0; st < ST; ++st) yacc[nt][st] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool more = ti + 1 < my_tiles;
#pragma unroll 1
        for (int c = 0; c < KT; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            block_sync();
            if (c + 1 < KT) issue(tile, 1, c + 1, (c + 1) & 1);
            else if (more) issue(tile + (int)gridDim.x, 0, 0, 0);
            const char* ws = smem + (c & 1) * C::SLOT;
            const char* ot = smem + C::OFF_O + c * (TS * 64);
            u32x4 bo[ST];
#pragma unroll
            for (int st = 0; st < ST; ++st) bo[st] = *reinterpret_cast<const u32x4*>(ot + (st * 16 + r16) * 64 + fsw);
#pragma unroll
            for (int nt = 0; nt < NW; ++nt) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(ws + ((w * NW + nt) * 16 + r16) * 64 + fsw);
#pragma unroll
                for (int st = 0; st < ST; ++st) yacc[nt][st] = X::mma(af, bo[st], yacc[nt][st]);
            }
        }
        // ---- epilogue 1: y rows staged through the ring slot the last chunk occupied (slot 1), stored 16 bytes per thread
        block_sync();                           // every wave is done reading slot 1
        {
            const float* bias = reinterpret_cast<const float*>(smem + C::OFF_BIAS) + E;
            char* ys = smem + C::SLOT;
#pragma unroll
            for (int nt = 0; nt < NW; ++nt) {
                const int n = ncol0 + 16 * nt + 4 * lg;
                const f32x4 bo4 = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                for (int st = 0; st < ST; ++st) {
                    const int row = st * 16 + r16;
                    *reinterpret_cast<u32x2*>(ys + row * C::YROW + n * 2) =
                        u32x2{pack_bf16x2(yacc[nt][st][0] + bo4[0], yacc[nt][st][1] + bo4[1]),
                              pack_bf16x2(yacc[nt][st][2] + bo4[2], yacc[nt][st][3] + bo4[3])};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            block_sync();
            for (int i = threadIdx.x; i < TS * (E / 8); i += 512) {
                const int row = i / (E / 8), ch = i - row * (E / 8);
                if (b0 + row < B)
                    *reinterpret_cast<u32x4*>(p.y + (b0 + row) * E + ch * 8) = *reinterpret_cast<const u32x4*>(ys + row * C::YROW + ch * 16);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NW, int M_, int ST>
void launch_row_fwd_t(const RowFwdArgs& a, hipStream_t s) {
    using C = RowFwdCfg<NW, M_, ST>;
    const int ntile = (int)((a.B + C::TS - 1) / C::TS);
    int grid = 256;
    if (grid > ntile) grid = ntile;
    auto kern = row_fwd_kernel<NW, M_, ST>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::TOTAL);
    kern<<<dim3(grid), dim3(512), C::TOTAL, s>>>(a, ntile);
}

}  // namespace

// shapes the fused forward takes: bf16, E in {256, 384, 512}, M <= 4, a wave's 16 NW columns inside one head
bool row_fwd_supported(int dtype, int E, int M, int H) {
    if (dtype != 0 || M < 1 || M > 4 || H < 1 || H > HPAD) return false;
    if (E != 256 && E != 384 && E != 512) return false;
    const int hd = E / H, nwc = E / 8;
    return hd * H == E && hd % nwc == 0;
}

void launch_row_fwd(const GateArgs& g, const GemmNtArgs& v, const GemmNtArgs& y, hipStream_t s) {
    RowFwdArgs a;
    a.x = (const unsigned short*)g.x; a.w_v = (const unsigned short*)v.w; a.b_v = (const unsigned short*)v.bias;
    a.w_o = (const unsigned short*)y.w; a.b_o = (const unsigned short*)y.bias;
    a.a_hi = (const unsigned short*)g.a_hi; a.a_lo = (const unsigned short*)g.a_lo; a.kpm = g.kpm; a.uniforms = g.uniforms; a.ph = g.ph;
    a.y = (unsigned short*)y.c; a.o = (unsigned short*)v.c; a.v = (unsigned short*)v.v_out; a.probs = g.probs;
    a.attn_w = g.attn_w; a.masked_w = g.masked_w; a.entropy = g.entropy; a.mask_rate = g.mask_rate;
    a.i_attn_w = (unsigned short*)g.i_attn_w; a.i_masked_w = (unsigned short*)g.i_masked_w;
    a.i_entropy = (unsigned short*)g.i_entropy; a.i_mask_rate = (unsigned short*)g.i_mask_rate;
    a.i_target = (unsigned short*)g.i_target; a.target_value = g.target_value;
    a.B = g.B; a.H = g.H; a.hd = g.E / g.H; a.mask = g.mask;
#define ROW_FWD_CASE(NW_)                                              \
    switch (g.M) {                                                     \
        case 1: launch_row_fwd_t<NW_, 1, 2>(a, s); break;              \
        case 2: launch_row_fwd_t<NW_, 2, 2>(a, s); break;              \
        case 3: launch_row_fwd_t<NW_, 3, 2>(a, s); break;              \
        default: launch_row_fwd_t<NW_, 4, 2>(a, s); break;             \
    }
    if (g.E == 256) { ROW_FWD_CASE(2) }
    else if (g.E == 384) { ROW_FWD_CASE(3) }
    else { ROW_FWD_CASE(4) }
#undef ROW_FWD_CASE
}

}  // namespace aecf
