// Weight-stationary streaming GEMM for the fusion pool's projections, bf16, gfx950:
//     C[r][n] = sum_k A[r][k] W[n][k] + bias[n]          R = B or B*M rows (huge), N, K <= 512 (tiny)
// A normal tiled GEMM re-stages the SAME small weight matrix for every row tile and spends its time in
// prologues, barriers and epilogues of 8-step K loops.  Here the weights never move:
//   * a block owns 256 output columns; each of its 8 waves keeps its 32 x K slice of W in REGISTERS as MFMA operands
//     (K = 512: 128 VGPRs) for the whole kernel -- loaded once;
//   * the block walks down its share of the rows; each step a [rows x K] tile of A is copied global -> LDS by the DMA
//     engine (global_load_lds, two buffers: the copy of step s+1 flies behind the MFMAs of step s), and every wave
//     multiplies the whole tile against its resident weights: 2 LDS operand reads feed 4 MFMAs;
//   * the product is formed transposed (weights as the MFMA A operand), so a lane ends up with 8 consecutive output
//     columns of ONE row: a 16-byte store, and -- for the value projection -- all M modality products of a sample in
//     the same lane, which makes the softmax-weighted sum o = sum_m p_m V_m a per-lane FMA.
// Modes:  PLAIN  C = A W^T + bias                                  (out-projection, dout = dy W_o)
//         VPROJ  rows (b,m) of x:  V[b,m,:] = x[b,m,:] W_v^T + b_v (saved for the backward, optional) and
//                o[b,n] = sum_m probs[b, head(n), m] V[b,m,n]
//         VFLAT  the same result for M = 4 when 16 samples x M rows x K do not fit LDS twice (K = 768 / 1024): the
//                (b,m) rows are walked as PLAIN rows (32 per step = 8 samples), a lane holds ONE (b,m) row, so its
//                softmax weight is a per-lane scalar and o is a sum over the 4 lanes of a quad (two xor shuffles)
// One barrier per step; LDS tile rows are K*2 bytes with the 16-byte chunk index XOR-ed with the MFMA column index
// (bank-conflict-free ds_read_b128; the DMA destination is lane-linear, so the XOR is applied to the source address).
#include <stdlib.h>

#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

namespace {

enum { WS_PLAIN = 0, WS_VPROJ = 1, WS_VFLAT = 2 };

// LDS operand reads are issued this many items ahead of their MFMAs (depths 2 and 5 measured level, profiles/r04_c2_*_ablation.txt)
constexpr int WS_PF = 3, VPROJ_PF = 3, DX_PF = 3;

// LDS tiles of gemm_ws_kernel: two.  (A third tile for the plain form -- copy two steps ahead -- measured neutral at C2, 38 us
// either way, profiles/r04_c2_plain_ablation.txt: copy, stores and MFMA + operand reads each take ~19 us on their own and
// already overlap.  The phase-ablation and in-kernel timeline builds these numbers came from are kept out of the product
// source: tools/micro/variants/.)
constexpr int ws_plain_bufs(int, int) { return 2; }

// copy NROWS rows (K bf16 each) to an LDS tile; row r's physical chunk p holds logical chunk p ^ key(r),
// key(r) = (r / KEYDIV) & 15.  Rows >= rows_valid re-read the last valid row (their outputs are never stored).
template <int KT, int NROWS, int KEYDIV>
__device__ __forceinline__ void ws_dma_rows(const char* __restrict__ src, unsigned int ld_bytes, int rows_valid, char* lds) {
    constexpr int CPR = 4 * KT;                       // 16-byte chunks per row
    constexpr int TOTAL = NROWS * CPR;
    constexpr int NI = (TOTAL + 511) / 512;
    static_assert(TOTAL % 64 == 0, "whole wave-instructions");
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (wbase + 512 * i < TOTAL) {                // wave-uniform
            const int c = threadIdx.x + 512 * i;
            const int row = c / CPR, p = c - row * CPR;
            const int rowc = row < rows_valid ? row : rows_valid - 1;
            const int key = (row / KEYDIV) & 15;
            const unsigned int voff = (unsigned)rowc * ld_bytes + (unsigned)((p ^ key) << 4);
            __builtin_amdgcn_global_load_lds(src + voff, (lds_void_t*)(lds + (wbase + 512 * i) * 16), 16, 0, 0);
        }
    }
}

// GATE (VPROJ only): the softmax weights are PRODUCED here instead of read -- scores x . A[h] against the folded key
// matrix (bf16 hi/lo split), one more use of the tile that is already in LDS, so the separate gate kernel and its pass
// over x disappear.  Each wave takes KG = KT/8 K-steps of the score product for all 16 heads x 16 samples x M
// (2 M KG MFMAs), the partial sums are folded through LDS in wave order, and every wave forms the softmax of ITS head for
// its lanes' samples (per-lane scalars in this layout).  The per-sample statistics (head mean, curriculum masking)
// are a tiny follow-up kernel on the saved weights (gate_stats_kernel).
// 4*CT consecutive bf16 output columns of one row: one 16-byte (CT = 2) or 8-byte (CT = 1) store
template <int CT>
__device__ __forceinline__ void store_packed(unsigned short* dst, const unsigned int* pk) {
    if (CT == 2) *reinterpret_cast<u32x4*>(dst) = u32x4{pk[0], pk[1], pk[2], pk[3]};
    else *reinterpret_cast<u32x2*>(dst) = u32x2{pk[0], pk[1]};
}
// the same columns as float32 (precise mode: intermediates and outputs are not rounded to bf16)
template <int CT>
__device__ __forceinline__ void store_cols_f32(float* dst, const float* v) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    if (CT == 2) *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
// the bf16 low parts of the same columns: v - float(bf16(v)), rounded to bf16
template <int CT>
__device__ __forceinline__ void store_cols(unsigned short* dst, const float* v);
template <int CT>
__device__ __forceinline__ void store_cols_lo(unsigned short* dst, const float* v) {
    float lo[4 * CT];
#pragma unroll
    for (int j = 0; j < 4 * CT; ++j) lo[j] = v[j] - Tr<BF16>::to_f32(Tr<BF16>::from_f32(v[j]));
    store_cols<CT>(dst, lo);
}
template <int CT>
__device__ __forceinline__ void store_cols(unsigned short* dst, const float* v) {
    if (CT == 2) {
        *reinterpret_cast<u32x4*>(dst) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                               pack_bf16x2(v[6], v[7])};
    } else {
        *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
}

// softmax over M_ scores in registers, for the gated value projections (bf16 path): v_exp_f32 on the base-2 scaled argument
// and one reciprocal with a Newton step instead of expf / IEEE division (each of which expands to ~12 instructions, three of
// each per step, in front of the MFMA phase every wave waits for).  Relative error of a weight ~1e-7 (v_exp_f32 is exact to
// 1 ulp in its range; masked scores are -inf -> exp2 gives 0).  An all-masked row (sum = 0) gives NaN like the reference's.
template <int M_>
__device__ __forceinline__ void softmax_fast(float* sc, float mx) {
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < M_; ++m) {
        sc[m] = __builtin_amdgcn_exp2f((sc[m] - mx) * 1.44269504088896340736f);
        sum += sc[m];
    }
    float inv = __builtin_amdgcn_rcpf(sum);
    inv = inv * (2.0f - sum * inv);
#pragma unroll
    for (int m = 0; m < M_; ++m) sc[m] *= inv;
}

// the NV consecutive bias values of a lane as ONE load (NV element-wise `bs ? bs[i] : 0` selects made hipcc emit NV branches,
// each with its own load and s_waitcnt vmcnt(0): eight dependent memory round trips in every launch's prologue)
template <int NV>
__device__ __forceinline__ void load_bias(const unsigned short* bs, int col, float* bias) {
#pragma unroll
    for (int j = 0; j < NV; ++j) bias[j] = 0.f;
    if (bs) {
        if (NV == 8) {
            Tr<BF16>::unpack(*reinterpret_cast<const u32x4*>(bs + col), bias);
        } else {
            Tr<BF16>::load4(bs + col, bias);
        }
    }
}

// CT = column tiles (of 16) per wave: 2 for K <= 512 (32 columns x K weights = up to 128 VGPRs, a block owns 256 columns),
// 1 for K = 768 / 1024 (16 columns, a block owns 128).
template <int KT, int MODE, int M_, bool GATE, int CT>
__global__ __launch_bounds__(512, 2) void gemm_ws_kernel(GemmNtArgs p, int rows_per_block, int nchunk) {
    using X = Tr<BF16>;
    constexpr int K = 32 * KT, ROWB = 2 * K;
    constexpr bool PL = MODE != WS_VPROJ;                         // plain row tiles (PLAIN, VFLAT)
    constexpr bool VF = MODE == WS_VFLAT;
    constexpr int NPM = MODE == WS_VPROJ ? M_ : (VF ? 2 : 1);     // softmax weights a lane needs per step
    constexpr int RT = PL ? 2 : M_;                 // 16-row MFMA tiles per step
    constexpr int SROWS = 16 * RT;                                // A rows per step
    constexpr int OROWS = PL ? 32 : 16;             // output rows (PLAIN) / samples (VPROJ) per step
    constexpr int TILE = SROWS * ROWB;
    constexpr int NBUF = ws_plain_bufs(MODE, KT);                 // PLAIN: three tiles, the copy runs TWO steps ahead
    constexpr int NI_DMA = SROWS * 4 * KT / 512;                  // LDS-DMA wave-instructions of one tile per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    const int N = p.N, H = p.H;
    // 1-D grid, XCD-aware: the column groups of one row chunk get ids b, b+8, ... -> same XCD, dispatched together,
    // so the chunk's rows come from HBM once and the other groups read them from that XCD's L2 (aecf_tile.h)
    unsigned int chunk_u, group_u;
    constexpr int CW = 16 * CT, BC = 8 * CW, NV = 4 * CT;        // columns per wave / per block, values per lane and row
    if (!xcd_tile(blockIdx.x, (unsigned)nchunk, (unsigned)(p.N / BC), chunk_u, group_u)) return;
    const int ncol0 = (int)group_u * BC + CW * w;                 // this wave's output columns
    // output rows (PLAIN) / samples (VPROJ) of this block
    const int64_t o_beg = (int64_t)chunk_u * rows_per_block;
    const int64_t o_all = p.R;                                    // rows (PLAIN, VFLAT: B*M of them) or samples (VPROJ)
    const int64_t o_end = (o_beg + rows_per_block) < o_all ? (o_beg + rows_per_block) : o_all;
    if (o_beg >= o_end) return;

    const char* asrc = reinterpret_cast<const char*>(p.a);
    // A rows: PLAIN row r of a [R, lda] matrix; VPROJ row (b, m) of x viewed as [B*M, K] (lda = M*K)
    const unsigned int row_pitch = (PL ? (unsigned)p.lda : (unsigned)K) * 2u;
    const int a_rows_per_o = PL ? 1 : M_;

    auto issue = [&](int64_t o0, int buf) {                        // DMA of the step that starts at output row o0
        const int ov = (int)((o_end - o0) < OROWS ? (o_end - o0) : OROWS);
        ws_dma_rows<KT, SROWS, (PL ? 1 : M_)>(asrc + o0 * a_rows_per_o * (int64_t)row_pitch, row_pitch,
                                                            ov * a_rows_per_o, smem + buf * TILE);
    };

    issue(o_beg, 0);                                              // the first tile flies while the weights load
    if (NBUF == 3 && o_beg + OROWS < o_end) issue(o_beg + OROWS, 1);

    // side job of the launch's first block (out-projection of a training-mode forward): the entropy regulariser's final sum
    // over the statistics kernel's per-block partials (ref aecf/AECFLayer.py:309-314: mean, clamp at 0) -- one wave, fixed order
    if (MODE == WS_PLAIN && !GATE && p.ent_loss && chunk_u == 0 && group_u == 0 && w == 0) {
        float acc = 0.f;
        for (int i = lane; i < p.ent_nblk; i += 64) acc += p.ent_partial[i];
        acc = reduce_wave(acc);
        if (lane == 0) reinterpret_cast<unsigned short*>(p.ent_loss)[0] = X::from_f32(fmaxf(acc * p.ent_inv_n, 0.f));
    }

    // ---- resident weights: MFMA A operand, row i = 4 lg' + r of tile c  <->  column ncol0 + 8 (i >> 2) + 4 c + (i & 3)
    //      (so that accumulator lane (lg, r16) holds columns ncol0 + 8 lg + 4 c + 0..3 of row/sample r16)
    const unsigned short* wsrc = reinterpret_cast<const unsigned short*>(p.w);
    u32x4 wreg[KT][CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = CT == 2 ? ncol0 + 8 * (r16 >> 2) + 4 * c + (r16 & 3) : ncol0 + r16;
        const unsigned short* wr = wsrc + (int64_t)n * K + 8 * lg;
        // fragment-major copy (prep kernel): 1 KB contiguous per wave-instruction instead of 16 x 64-byte pieces
        const u32x4* wf = reinterpret_cast<const u32x4*>(p.w_frag) + ((int64_t)((ncol0 / CW) * CT + c) * KT) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) wreg[ks][c] = p.w_frag ? wf[ks * 64] : *reinterpret_cast<const u32x4*>(wr + 32 * ks);
    }
    float bias[NV];
    load_bias<NV>(reinterpret_cast<const unsigned short*>(p.bias), ncol0 + NV * lg, bias);
    const int head = ncol0 / p.hd;                                // VPROJ: the wave's 32 columns lie in one head

    // ---- LDS operand addresses: row (tile t: 16 t + r16 | modality m: r16 M + m), chunk (4 ks + lg) ^ r16
    int xaddr[4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
        xaddr[v] = (PL ? r16 : r16 * M_) * ROWB + ((((4 * v) + lg) ^ r16) << 4);

    // Synchronisation: the DMA of step s+1 is issued right after the barrier of step s and retired by the
    // s_waitcnt vmcnt(0) that FOLLOWS the MFMAs of step s (a whole step of compute later: it also retires the
    // previous step's stores, so the count is exact); this step's stores are issued after that wait and fly
    // behind the next step.  Reads of a buffer happen one barrier after the wait that retired its DMA.
    // VPROJ: this lane's sample, softmax weights of the wave's head, loaded ONE STEP AHEAD by inline asm (an ordinary
    // load beside an in-flight LDS-DMA makes hipcc wait vmcnt(0) in front of the first MFMA, which would serialise
    // the DMA with the compute); the end-of-step s_waitcnt vmcnt(0) retires them with the tile they belong to.
    float pm[NPM], pm_next[NPM];
    // GATE: this wave's K-steps of the score product and its operands (rows of A hi/lo = heads), partial-sum buffer
    constexpr int KG = (KT + 7) / 8;
    float* gpart = reinterpret_cast<float*>(smem + 2 * TILE);     // [8 waves][M][16 heads][16 samples]
    u32x4 ga[GATE ? KG : 1][2];
    if (GATE) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int ks = KG * w + kg;                           // waves beyond KT/KG own nothing: zero operands
            const unsigned short* ah = reinterpret_cast<const unsigned short*>(p.g_ahi) + (int64_t)r16 * K + 32 * ks + 8 * lg;
            const unsigned short* al = reinterpret_cast<const unsigned short*>(p.g_alo) + (int64_t)r16 * K + 32 * ks + 8 * lg;
            ga[kg][0] = ks < KT ? *reinterpret_cast<const u32x4*>(ah) : u32x4{0u, 0u, 0u, 0u};
            ga[kg][1] = ks < KT ? *reinterpret_cast<const u32x4*>(al) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // key_padding_mask bytes of this lane's sample, fetched one step ahead like the probabilities used to be
    unsigned int kp[GATE ? M_ : 1], kp_next[GATE ? M_ : 1];
    auto load_probs = [&](int64_t o0, float* dst) {
        if (MODE == WS_VPROJ && GATE) {
            if (p.g_kpm) {
                const int64_t b = (o0 + r16) < o_end ? (o0 + r16) : (o_end - 1);
                const uint8_t* kq = p.g_kpm + b * M_;
#pragma unroll
                for (int m = 0; m < M_; ++m)
                    asm volatile("global_load_ubyte %0, %1, off" : "=v"(kp_next[m]) : "v"(kq + m) : "memory");
            }
        } else if (MODE == WS_VPROJ) {
            const int64_t b = (o0 + r16) < o_end ? (o0 + r16) : (o_end - 1);
            const float* pp = p.probs + (b * H + head) * M_;
#pragma unroll
            for (int m = 0; m < M_; ++m)
                asm volatile("global_load_dword %0, %1, off" : "=v"(dst[m]) : "v"(pp + m) : "memory");
        } else if (VF) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {                         // this lane's two (b, m) rows of the step
                const int64_t row = (o0 + 16 * t + r16) < o_end ? (o0 + 16 * t + r16) : (o_end - 1);
                const int64_t b = row / M_;
                const float* pp = p.probs + (b * H + head) * M_ + (row - b * M_);
                asm volatile("global_load_dword %0, %1, off" : "=v"(dst[t]) : "v"(pp) : "memory");
            }
        }
    };
    load_probs(o_beg, pm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < NPM; ++m) {
        if (GATE) { asm volatile("" : "+v"(kp_next[m])); kp[m] = p.g_kpm ? kp_next[m] : 0u; }
        else asm volatile("" : "+v"(pm[m]));
    }
    int cur = 0;
    for (int64_t o0 = o_beg; o0 < o_end; o0 += OROWS, cur = (cur + 1 == NBUF ? 0 : cur + 1)) {
        __builtin_amdgcn_s_barrier();                              // tile visible to all waves; other buffer free
        // GATE: the next tile's copy is issued among the first MFMAs of the step (below), not here: at the top its six
        // wave-instructions and their address arithmetic sit in front of the score phase, which every wave of the block
        // then waits for at the second barrier (value projection 130 -> 121 us at C2)
        // PLAIN (three buffers): the copy of step s + 2 goes into the tile step s - 1 read (every wave is past it: barrier);
        // a step of 64 MFMAs per wave is shorter than one HBM round trip under load, so one step of lead left the wait at the
        // end of every step exposed
        const bool two_ahead = NBUF == 3 && o0 + 2 * OROWS < o_end;
        if (NBUF == 3) { if (two_ahead) issue(o0 + 2 * OROWS, cur >= 1 ? cur - 1 : 2); }
        else if (!GATE && o0 + OROWS < o_end) issue(o0 + OROWS, cur ^ 1);
        if (o0 + OROWS < o_end) load_probs(o0 + OROWS, pm_next);

        if (GATE) {
            // ---- scores of this step: partial product over this wave's K-steps, all heads x 16 samples x M
            const char* tg = smem + cur * TILE;
            f32x4 gacc[M_];
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                gacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
                    const int ks = (KG * w + kg) % KT;             // (wave-uniform; owners only contribute non-zero)
                    const int xo = r16 * M_ * ROWB + m * ROWB + ((((ks & 3) * 4 + lg) ^ r16) << 4) + (ks >> 2) * 256;
                    const u32x4 xf = *reinterpret_cast<const u32x4*>(tg + xo);
                    gacc[m] = X::mma(ga[kg][0], xf, gacc[m]);
                    gacc[m] = X::mma(ga[kg][1], xf, gacc[m]);
                }
                // accumulator lane (lg, r16), register r: head 4 lg + r, sample r16
#pragma unroll
                for (int r = 0; r < 4; ++r) gpart[((w * M_ + m) * 16 + 4 * lg + r) * 16 + r16] = gacc[m][r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // softmax over the modalities for (this wave's head, this lane's sample); partial sums in wave order
            float sc[M_], mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                float a = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) a += gpart[((ww * M_ + m) * 16 + head) * 16 + r16];
                if (kp[m] != 0u) a = -INFINITY;                   // torch functional.py:6554-6566
                sc[m] = a;
                mx = fmaxf(mx, a);
            }
            softmax_fast<M_>(sc, mx);
            const int64_t bq = o0 + r16;
            const bool writer = lg == 0 && ncol0 % p.hd == 0 && bq < o_end;         // first wave of the head, one lane per sample
#pragma unroll
            for (int m = 0; m < M_; ++m) pm[m] = sc[m];
            if (writer) {
                float* dst = const_cast<float*>(p.probs) + (bq * H + head) * M_;
#pragma unroll
                for (int m = 0; m < M_; ++m) dst[m] = pm[m];
            }
        }

        // ---- products.  One "item" = one LDS operand read + the 2 MFMAs it feeds (the wave's two 16-column tiles).
        //      PLAIN: items run k-major over the two row tiles (4 independent accumulators); VPROJ: modality-major,
        //      so only one modality's accumulators are live and V_m / o are finished as each modality completes.
        //      The operand reads run PF items ahead of the MFMAs (a ring of PF+1 registers sets, pinned with
        //      sched_group_barrier so the compiler keeps the reads early instead of sinking them next to their use).
        constexpr int NIT = RT * KT;
        constexpr int PF = WS_PF;
        const char* tb = smem + cur * TILE;
        auto rd = [&](int i) -> u32x4 {
            const int ks = PL ? i / RT : i % KT;
            const int t = PL ? i % RT : i / KT;
            const int toff = (PL ? 16 * t * ROWB : t * ROWB) + (ks >> 2) * 256;
            return *reinterpret_cast<const u32x4*>(tb + xaddr[ks & 3] + toff);
        };
        constexpr int NACC = PL ? RT : 1;
        f32x4 acc[NACC][CT];
#pragma unroll
        for (int t = 0; t < NACC; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float ov[NV];
        unsigned int vpk[MODE == WS_VPROJ ? M_ : 1][NV / 2];      // V_m + bias, packed bf16 pairs
#pragma unroll
        for (int j = 0; j < NV; ++j) ov[j] = 0.f;

        u32x4 xf[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; ++i) xf[i] = rd(i);
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);        // the PF reads of the ring's head go first
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (i + PF < NIT) xf[(i + PF) % (PF + 1)] = rd(i + PF);
            if (GATE && i == NIT / 8 && o0 + OROWS < o_end) issue(o0 + OROWS, cur ^ 1);   // 7/8 of the MFMA phase to land in
            const int ks = PL ? i / RT : i % KT;
            const int t = PL ? i % RT : 0;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[t][c] = X::mma(wreg[ks][c], xf[i % (PF + 1)], acc[t][c]);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);    // CT MFMA
            if (MODE == WS_VPROJ && ks == KT - 1) {                // modality m = i / KT is complete
                const int m = i / KT;
                float v[NV];
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[4 * c + r] = acc[0][c][r] + bias[4 * c + r];
                        ov[4 * c + r] = fmaf(pm[m], v[4 * c + r], ov[4 * c + r]);
                    }
#pragma unroll
                for (int j = 0; j < NV / 2; ++j) vpk[m][j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[0][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }

        // next tile (and its probabilities) landed.  Three buffers: only the NI_DMA copies of step s + 2 -- the youngest loads --
        // may stay in flight (loads return in order, so a count <= NI_DMA cannot include a copy of step s + 1; the previous
        // step's stores are older than those copies and are retired here as well, whatever order they complete in)
        if (two_ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI_DMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float pm_now[NPM];                                         // (the weights of THIS step: pm is refilled below)
#pragma unroll
        for (int m = 0; m < NPM; ++m) pm_now[m] = pm[m];
#pragma unroll
        for (int m = 0; m < NPM; ++m) {
            if (GATE) {
                asm volatile("" : "+v"(kp_next[m]));
                kp[m] = p.g_kpm ? kp_next[m] : 0u;
            } else {
                asm volatile("" : "+v"(pm_next[m]));
                pm[m] = pm_next[m];
            }
        }
        // ---- stores: lane (lg, r16) holds columns ncol0 + NV lg + 4 c + r (c < CT; r = 0..3) of row / sample r16
        if (PL) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int64_t row = o0 + 16 * t + r16;
                float v[NV];
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * c + r] = acc[t][c][r] + bias[4 * c + r];
                if (!VF) {
                    if (row < o_end) {
                        if (p.out_f32) store_cols_f32<CT>(reinterpret_cast<float*>(p.c) + row * N + ncol0 + NV * lg, v);
                        else store_cols<CT>(reinterpret_cast<unsigned short*>(p.c) + row * N + ncol0 + NV * lg, v);
                        if (p.c_lo) store_cols_lo<CT>(reinterpret_cast<unsigned short*>(p.c_lo) + row * N + ncol0 + NV * lg, v);
                    }
                } else {
                    if (p.v_out && row < o_end)
                        store_cols<CT>(reinterpret_cast<unsigned short*>(p.v_out) + row * N + ncol0 + NV * lg, v);
                    float ow[NV];
#pragma unroll
                    for (int j = 0; j < NV; ++j) {                 // o = sum over the quad's 4 modality rows (M_ == 4)
                        float a = pm_now[t] * v[j];
                        a += __shfl_xor(a, 1, 64);
                        a += __shfl_xor(a, 2, 64);
                        ow[j] = a;
                    }
                    if ((r16 & 3) == 0 && row < o_end)
                        store_cols<CT>(reinterpret_cast<unsigned short*>(p.c) + (row / M_) * N + ncol0 + NV * lg, ow);
                }
            }
        } else {
            const int64_t b = o0 + r16;
            if (b < o_end) {
                if (p.v_out) {
#pragma unroll
                    for (int m = 0; m < M_; ++m)
                        store_packed<CT>(reinterpret_cast<unsigned short*>(p.v_out) + (b * M_ + m) * N + ncol0 + NV * lg, vpk[m]);
                }
                if (p.out_f32) store_cols_f32<CT>(reinterpret_cast<float*>(p.c) + b * N + ncol0 + NV * lg, ov);
                else store_cols<CT>(reinterpret_cast<unsigned short*>(p.c) + b * N + ncol0 + NV * lg, ov);
                if (p.c_lo) store_cols_lo<CT>(reinterpret_cast<unsigned short*>(p.c_lo) + b * N + ncol0 + NV * lg, ov);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dx on the same engine: the softmax weights applied on the OUTPUT side, per head, in the accumulator layout.
// (A first form scaled M copies of each do row on the input side -- a cooperative pass into a second LDS tile, two
// barriers per step, 3x the operand reads: 122 us against 87 us for this one at C2.)
//     P_h[b, k] = sum_{j in head h} do[b, j] W_v[j, k]            (MFMA over the head's HK = hd/32 K-steps)
//     dx[b, m, k] = sum_h probs[b, h, m] P_h[b, k]  +  sum_h ds[b, h, m] A[h, k]
// In the transposed product a lane holds 8 output columns of ONE sample, so probs[b, h, m] is a per-lane scalar and
// the weighting is M*8 FMAs per head on registers: no scaled operand tile, no scaling pass, no second barrier, and the
// LDS operand traffic of one raw do row per sample instead of M scaled rows.  The key-side term is KX extra MFMA
// K-steps whose B operand ([ds_hi | ds_lo | ds_hi] entries of the lane's sample) is built in registers.
template <int KT, int HK, int M_, int CT>
__global__ __launch_bounds__(512, 2) void dx_ws2_kernel(BwdGArgs p, int rows_per_block, int nchunk) {
    using X = Tr<BF16>;
    constexpr int CW = 16 * CT, BC = 8 * CW, NV = 4 * CT;
    constexpr int K = 32 * KT, ROWB = 2 * K;
    constexpr int H_ = KT / HK;                                    // heads
    constexpr int KX = (3 * H_ + 31) / 32;
    constexpr int RAW = 16 * ROWB;
    constexpr int HM = H_ * M_, NST = 16 * HM;
    constexpr int HMS = HM | 1;                                    // stage row stride in floats: odd, so the 16 samples' reads of one (h, m) hit 16 banks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XROWB = 64 * KX + 16;                           // bytes per row of the key-side operand tile (+16: the samples' rows start in different banks)
    constexpr int XT = 16 * M_ * XROWB;
    char* raw = smem;                                             // [2][16][K] do rows, chunk ^ row
    float* stage = reinterpret_cast<float*>(smem + 2 * RAW);      // [2][16][H][M] softmax weights, double buffered
    char* xtra = smem + 2 * RAW + 2 * 16 * HMS * 4;               // [2][16*M][32*KX (+8)] bf16 [ds_hi | ds_lo | ds_hi] rows

    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    const int E = p.E;
    // side job, every block of the launch (before any of them leaves): u[h][k] = sum of the u slabs the score-gradient kernel
    // wrote, elements dealt evenly over the blocks, threads = (element, slab group), groups folded in fixed order through
    // LDS (the second do buffer: nothing lands there before the first step) -- the finalize launch then reads 16 KB of
    // reduced u instead of depending on a reduction launch of its own
    if (p.u_slab_in) {
        const int n = H_ * E, G = (int)gridDim.x;
        const int per = (n + G - 1) / G, e0 = (int)blockIdx.x * per;
        const int ew = (n - e0) < per ? (n - e0) : per;
        if (ew > 0) {
            int ewp = 1;
            while (ewp < ew && ewp < 512) ewp <<= 1;
            float* scratch = reinterpret_cast<float*>(raw + RAW);
            for (int base = 0; base < ew; base += ewp) {         // (ew <= 512 whenever the grid has >= n / 512 blocks)
                const int el = base + (int)(threadIdx.x & (ewp - 1)), sl = (int)threadIdx.x / ewp, nsl = 512 / ewp;
                float acc = 0.f;
                if (el < ew)
                    for (int sb = sl; sb < p.u_nslab; sb += nsl) acc += p.u_slab_in[(int64_t)sb * n + e0 + el];
                scratch[threadIdx.x] = acc;
                __syncthreads();
                if ((int)threadIdx.x < ewp && base + (int)threadIdx.x < ew) {
                    float t = 0.f;
                    for (int g = 0; g < nsl; ++g) t += scratch[g * ewp + threadIdx.x];
                    p.u_out[e0 + base + threadIdx.x] = t;
                }
                __syncthreads();
            }
        }
    }
    unsigned int chunk_u, group_u;
    if (!xcd_tile(blockIdx.x, (unsigned)nchunk, (unsigned)(E / BC), chunk_u, group_u)) return;
    const int ncol0 = (int)group_u * BC + CW * w;
    const int64_t o_beg = (int64_t)chunk_u * rows_per_block;
    const int64_t o_end = (o_beg + rows_per_block) < p.B ? (o_beg + rows_per_block) : p.B;
    if (o_beg >= o_end) return;

    const char* dsrc = reinterpret_cast<const char*>(p.dobuf);
    auto issue = [&](int64_t o0, int buf) {
        const int ov = (int)((o_end - o0) < 16 ? (o_end - o0) : 16);
        ws_dma_rows<KT, 16, 1>(dsrc + o0 * (int64_t)ROWB, (unsigned)ROWB, ov, raw + buf * RAW);
    };
    issue(o_beg, 0);                                              // the first rows fly while the weights load

    const unsigned short* wsrc = reinterpret_cast<const unsigned short*>(p.wvt);
    u32x4 wreg[KT + KX][CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = CT == 2 ? ncol0 + 8 * (r16 >> 2) + 4 * c + (r16 & 3) : ncol0 + r16;
        const unsigned short* wr = wsrc + (int64_t)n * K + 8 * lg;
        const u32x4* wf = reinterpret_cast<const u32x4*>(p.wvt_frag) + ((int64_t)((ncol0 / CW) * CT + c) * KT) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) wreg[ks][c] = p.wvt_frag ? wf[ks * 64] : *reinterpret_cast<const u32x4*>(wr + 32 * ks);
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) {
            // (all eight loads first, from always-valid addresses, THEN the selects: a load inside `if (e < 3 H)` -- a lane-
            // dependent condition -- made hipcc branch around each one and wait vmcnt(0) behind it: 16 dependent memory round
            // trips in this prologue, most of the 17 us this kernel cost per launch whatever the batch)
            float raw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 32 * kx + 8 * lg + j;               // entry: [0,H) A_hi, [H,2H) A_hi, [2H,3H) A_lo
                const int h = (e < 3 * H_ ? e : 0) % H_;
                raw[j] = p.a_f32[(int64_t)h * E + n];
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 32 * kx + 8 * lg + j;
                const float hi = X::to_f32(X::from_f32(raw[j]));
                v[j] = e < 2 * H_ ? hi : (e < 3 * H_ ? raw[j] - hi : 0.f);
            }
            wreg[KT + kx][c] = X::pack(v);
        }
    }
    int xaddr[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) xaddr[v] = r16 * ROWB + ((((4 * v) + lg) ^ r16) << 4);

    constexpr int NSI = (NST + 511) / 512;
    float stg[NSI][2];
    auto load_stage = [&](int64_t o0) {
#pragma unroll
        for (int i = 0; i < NSI; ++i) {
            const int idx = threadIdx.x + 512 * i;
            const int bb = idx / HM, rem = idx - bb * HM;
            const int64_t b = (o0 + bb) < o_end ? (o0 + bb) : (o_end - 1);
            const float* pp = p.probs + b * HM + rem;
            const float* dp = p.dsbuf + b * HM + rem;
            if (idx < NST) {
                asm volatile("global_load_dword %0, %1, off" : "=v"(stg[i][0]) : "v"(pp) : "memory");
                asm volatile("global_load_dword %0, %1, off" : "=v"(stg[i][1]) : "v"(dp) : "memory");
            }
        }
    };
    auto park_stage = [&](int buf) {                               // into the buffer the NEXT step reads
#pragma unroll
        for (int i = 0; i < NSI; ++i) {
            const int idx = threadIdx.x + 512 * i;
            asm volatile("" : "+v"(stg[i][0]), "+v"(stg[i][1]));
            if (idx < NST) {
                stage[buf * 16 * HMS + (idx / HM) * HMS + idx % HM] = stg[i][0];
                const int bb = idx / HM, rem = idx - bb * HM, h = rem / M_, m = rem - h * M_;
                const float d = stg[i][1];
                const unsigned short hi = X::from_f32(d);
                const unsigned short lo = X::from_f32(d - X::to_f32(hi));
                unsigned short* xr = reinterpret_cast<unsigned short*>(xtra + buf * XT + (bb * M_ + m) * XROWB);
                xr[h] = hi;
                xr[H_ + h] = lo;
                xr[2 * H_ + h] = hi;
            }
        }
    };

    for (int i = threadIdx.x; i < 2 * XT / 4; i += 512) reinterpret_cast<unsigned int*>(xtra)[i] = 0u;   // padding entries
    load_stage(o_beg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    park_stage(0);
    int cur = 0;
    for (int64_t o0 = o_beg; o0 < o_end; o0 += 16, cur ^= 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // do rows + stage of this step visible
        if (o0 + 16 < o_end) load_stage(o0 + 16);                  // (the rows' copy is issued behind the first head's MFMAs, below)
        const char* tb = raw + cur * RAW;
        const float* sp = stage + cur * 16 * HMS + r16 * HMS;      // this lane's sample: probs[h][m]
        const char* xq = xtra + cur * XT + r16 * M_ * XROWB + 16 * lg;
        f32x4 acc[M_][CT];
#pragma unroll
        for (int m = 0; m < M_; ++m)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};

        // key-side term first: B operand of modality m = entries [ds_hi | ds_lo | ds_hi] of this lane's sample
#pragma unroll
        for (int m = 0; m < M_; ++m)
#pragma unroll
            for (int kx = 0; kx < KX; ++kx) {
                const u32x4 df = *reinterpret_cast<const u32x4*>(xq + m * XROWB + 64 * kx);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[m][c] = X::mma(wreg[KT + kx][c], df, acc[m][c]);
            }

        // value-side term, head by head; operand reads run PF K-steps ahead
        constexpr int PF = DX_PF;
        auto rd = [&](int ks) -> u32x4 {
            return *reinterpret_cast<const u32x4*>(tb + xaddr[ks & 3] + (ks >> 2) * 256);
        };
        u32x4 xf[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; ++i) xf[i] = rd(i);
        // software pipeline over heads: the MFMAs of head h+1 are issued in the same scheduling region as the FMAs that
        // apply head h (the matrix pipe and the vector ALU overlap); a scheduling fence per head keeps only two P sets live
        f32x4 Pc[CT], Pn[CT];
        auto head_mma = [&](int h, f32x4* P) {
#pragma unroll
            for (int c = 0; c < CT; ++c) P[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kl = 0; kl < HK; ++kl) {
                const int ks = h * HK + kl;
                if (ks + PF < KT) xf[(ks + PF) % (PF + 1)] = rd(ks + PF);
#pragma unroll
                for (int c = 0; c < CT; ++c) P[c] = X::mma(wreg[ks][c], xf[ks % (PF + 1)], P[c]);
            }
        };
        head_mma(0, Pc);
        if (o0 + 16 < o_end) issue(o0 + 16, cur ^ 1);
#pragma unroll
        for (int h = 0; h < H_; ++h) {
            if (h + 1 < H_) head_mma(h + 1, Pn);
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                const float a = sp[h * M_ + m];
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[m][c][r] = fmaf(a, Pc[c][r], acc[m][c][r]);
            }
            // keep the weighting of head h HERE (instruction sinking would otherwise move all H*M*8 FMAs behind the last
            // MFMA and keep every head's P live)
#pragma unroll
            for (int m = 0; m < M_; ++m)
#pragma unroll
                for (int c = 0; c < CT; ++c) asm volatile("" : "+v"(acc[m][c]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CT; ++c) Pc[c] = Pn[c];
        }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // next do rows + stage values landed
        if (o0 + 16 < o_end) park_stage(cur ^ 1);                  // the other buffer: nobody reads it during this step
        // (round 4, measured and dropped: the step's stores -- 16 of this kernel's 78 us by ablation, profiles/r04_c2_dx_ablation.txt --
        //  packed and issued one at a time behind the NEXT step's first heads' MFMAs, with and without a mask branch around
        //  them: 91-94 us against 76-80 -- a store inside the head pipeline costs more than eight waves storing at once here)
        const int64_t b = o0 + r16;
        if (b < o_end) {
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                float v[NV];
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * c + r] = acc[m][c][r];
                store_cols<CT>(reinterpret_cast<unsigned short*>(p.dx) + (b * M_ + m) * E + ncol0 + NV * lg, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Score gradient AND the key-side batch reduction from x, on the same engine -- no saved V, no separate pass over x:
//     P_h[b, k]  = sum_{j in head h} do[b, j] W_v[j, k]                       (as in dx_ws2, MFMA)
//     da[b,h,m]  = P_h[b, :] . x[b, m, :]            (= do_h . V_h[b,m] up to a term that cancels in the softmax backward)
//     ds[b,h,:]  = a (dp - sum_m a dp),  dp = da + dwbar/H                     (dscore_v_kernel's arithmetic)
//     u[h, k]   += sum_{b,m} ds[b,h,m] x[b,m,k]                                (MFMA on TRANSPOSED reads of the x tile)
// The split over blocks is by HEADS, not by output columns: a block owns JB = 32 KJ columns j of do (whole heads) and ALL
// E columns k, so the dot over k is complete inside the block (no cross-block reduction) and ds is final when the
// u product needs it.  Resident weights: W_v^T rows k = 16 NCT per wave, K = the block's JB columns (E = 512, JB = 256:
// 128 registers, as everywhere on this engine).  Per 16-sample step the x tile (16 M rows x E, LDS-DMA, two buffers) is
// used three times without leaving the CU: as the vector operand of the dot, and -- read column-wise with
// ds_read_b64_tr_b16 -- as the MFMA A operand of u^T = x^T ds (ds split hi/lo bf16).  HBM: do once, x once (the second
// head group of a row chunk finds the rows in its XCD's L2), ds written (tiny); the forward no longer stores V.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float dot2_bf16(unsigned int a, unsigned int b, float c) {     // v_dot2c_f32_bf16
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// WIDE (round 4): the resident weights' rows are permuted so that a lane's accumulators of a tile PAIR hold 8 CONSECUTIVE
// columns k (k = ncol0 + 32 (ct >> 1) + 8 lg + 4 (ct & 1) + reg, the trick of the plain kernel's 16-byte stores), so the dot
// reads x in 16-byte pieces: half the LDS reads of the 8-byte form -- each of them is waited for right in front of its use
// (the kernel has no registers left to run them ahead: profiles/r01_pmc_notes.md, end of round 3).
template <int KT, int KJ, int HK, int M_, bool PKDOT, int VAR, bool HILO = false>
__global__ __launch_bounds__(512, 2) void dsu_ws_kernel(BwdGArgs p, float* __restrict__ u_slab, int rows_per_block, int nchunk) {
    using X = Tr<BF16>;
    constexpr bool WIDE = VAR >= 1;                               // 16-byte dot reads (VAR 0: the 8-byte form of rounds 2-3)
    constexpr bool NOBR = VAR >= 2;                               // partial dots leave the wave without a branch (below)
    constexpr int E = 32 * KT, JB = 32 * KJ, NCT = KT / 4;        // a wave owns E / 8 = 16 NCT columns k
    constexpr int HBL = KJ / HK;                                  // heads per block
    constexpr int ROWX = 2 * E, ROWD = 2 * JB;
    constexpr int XROWS = 16 * M_;                                // x tile rows (b, m)
    constexpr int KU = (XROWS + 31) / 32;                         // MFMA K-steps of the u product (K = x tile row)
    constexpr int ZROWS = (32 * KU - XROWS) > 0 ? 1 : 0;          // K padding: one all-zero row (every padding slot reads it)
    constexpr int PSTR = HBL * M_ * 16;                           // floats per partial block (one per wave)
    constexpr int DSROW = 72;                                     // ds operand row (64 K slots + 8): rows start 9 x 16 B apart
    constexpr int XT = XROWS * ROWX, DT = 16 * ROWD;
    constexpr int NPART = 8;                                      // partial dots per (sample, head, m): one per wave (its four
                                                                  // lane groups are added in registers: v_permlane16/32_swap)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xb = smem;                                              // [2][16 M][E] x rows (b, m), 16-byte chunk ^ sample
    char* zp = smem + 2 * XT;                                     // [ZROWS][E] zeros
    char* db = zp + ZROWS * ROWX;                                 // [2][16][JB] do rows, chunk ^ row
    char* dbl = db + 2 * DT;                                      // HILO: [2][16][JB] low parts of the do rows
    float* part = reinterpret_cast<float*>(db + (HILO ? 4 : 2) * DT);   // [32][HM][16 samples]
    unsigned short* dsh = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(part) + NPART * PSTR * 4);     // [16][DSROW] bf16 hi
    unsigned short* dsl = dsh + 16 * DSROW;                                                                          // [16][DSROW] bf16 lo

    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int H = p.H;
    unsigned int chunk_u, group_u;
    if (!xcd_tile(blockIdx.x, (unsigned)nchunk, (unsigned)(E / JB), chunk_u, group_u)) return;
    const int jbase = (int)group_u * JB, hbase = jbase / (32 * HK);
    const int ncol0 = 16 * NCT * w;
    const int64_t o_beg = (int64_t)chunk_u * rows_per_block;
    const int64_t o_end = (o_beg + rows_per_block) < p.B ? (o_beg + rows_per_block) : p.B;
    if (o_beg >= o_end) return;

    const char* dsrc = reinterpret_cast<const char*>(p.dobuf) + (int64_t)jbase * 2;
    const char* xsrc = reinterpret_cast<const char*>(p.x);
    const char* dlsrc = HILO ? reinterpret_cast<const char*>(p.do_lo) + (int64_t)jbase * 2 : dsrc;
    auto issue = [&](int64_t o0, int buf) {
        const int ov = (int)((o_end - o0) < 16 ? (o_end - o0) : 16);
        ws_dma_rows_asm<KT, 16 * M_, M_>(xsrc + o0 * M_ * (int64_t)ROWX, (unsigned)ROWX, ov * M_, xb + buf * XT);
        ws_dma_rows_asm<KJ, 16, 1>(dsrc + o0 * (int64_t)ROWX, (unsigned)ROWX, ov, db + buf * DT);
        if (HILO) ws_dma_rows_asm<KJ, 16, 1>(dlsrc + o0 * (int64_t)ROWX, (unsigned)ROWX, ov, dbl + buf * DT);
    };
    // what the DMA never writes and the MFMAs still read: the zero page, the ds operand arrays
    for (int i = threadIdx.x; i < ZROWS * ROWX / 16; i += 512) reinterpret_cast<u32x4*>(zp)[i] = u32x4{0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < 2 * 16 * DSROW / 2; i += 512) reinterpret_cast<unsigned int*>(dsh)[i] = 0u;
    issue(o_beg, 0);

    // ---- resident weights: A operand row r16 of column tile ct <-> k = ncol0 + 16 ct + r16, K = j
    const unsigned short* wsrc = reinterpret_cast<const unsigned short*>(p.wvt);
    u32x4 wreg[KJ][NCT];
    static_assert(!WIDE || NCT % 2 == 0, "tile pairs");
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int krow = WIDE ? ncol0 + 32 * (ct >> 1) + 8 * (r16 >> 2) + 4 * (ct & 1) + (r16 & 3) : ncol0 + 16 * ct + r16;
        const unsigned short* wr = wsrc + (int64_t)krow * E + jbase + 8 * lg;
#pragma unroll
        for (int ks = 0; ks < KJ; ++ks) wreg[ks][ct] = *reinterpret_cast<const u32x4*>(wr + 32 * ks);
    }
    int daddr[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) daddr[v] = r16 * ROWD + ((((4 * v) + lg) ^ r16) << 4);
    // the 8-byte piece of x row (r16, m = 0) that holds columns ncol0 + 4 lg .. + 3 of column tile 0; tile ct adds
    // 2 ct to the chunk index BEFORE the xor with the sample, which (2 ct < 8, ncol0 / 8 a multiple of 8) is an xor of the
    // byte offset with 32 ct
    // WIDE: the 16-byte piece that holds columns ncol0 + 8 lg .. + 7 of tile pair 0; pair c2 adds 4 to the chunk index (bit 2 of
    // (ncol0 >> 3) + lg is clear), an xor of the byte offset with 64.  ds_read_b128's 16-lane groups then touch 16 different
    // 16-byte slots of the 256-byte bank window (lane-group rows {0-3,12-15 | 4-11} xor two chunk numbers one apart): conflict-free.
    const int xaddr0 = WIDE ? r16 * M_ * ROWX + ((((ncol0 >> 3) + lg) ^ r16) << 4)
                            : r16 * M_ * ROWX + (((((ncol0 >> 3) + (lg >> 1))) ^ r16) << 4) + 8 * (lg & 1);
    // transposed-read addresses of the u product (constant over the steps).  The K index of that product is any
    // numbering of the tile's (sample, modality) rows, as long as the ds operand arrays use the same one -- and the
    // row stride is a multiple of the 256-byte bank window, so rows read together must differ in their swizzle key
    // (= sample) by more than its low bit.  K slot s = 32 ks + 8 lg + 4 hh + q: the 8 rows a 32-lane half of one
    // transposed read touches (lg & 1, q) are the samples b = 2 j + parity, j = 4 (lg & 1) + q, of ONE modality m, with
    // (m, parity) = the "domain" D = 4 ks + 2 hh + (lg >> 1) = 2 m + parity; domains >= 2 M read the zero page.
    // (Rows in tile order -- 8 consecutive (b, m) rows per half -- shared keys three ways: SQ_LDS_BANK_CONFLICT was half
    // of SQ_LDS_IDX_ACTIVE for this kernel.)
    const int q = r16 >> 2, pp = r16 & 3;
    int ua[KU][2];
    bool uz[KU][2];
#pragma unroll
    for (int ks = 0; ks < KU; ++ks) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int D = 4 * ks + 2 * hh + (lg >> 1);
            uz[ks][hh] = D >= 2 * M_;
            const int m = D >> 1, b = 2 * (4 * (lg & 1) + q) + (D & 1);
            const int key = uz[ks][hh] ? 0 : b;
            const int ch = ((ncol0 >> 3) + (pp >> 1)) ^ key;
            ua[ks][hh] = (uz[ks][hh] ? (int)(zp - smem) : (b * M_ + m) * ROWX) + (ch << 4) + 8 * (pp & 1);
        }
    }
    // K slot of (sample b, modality m) in that numbering (the ds threads write their operand entries there)
    auto kslot = [&](int b, int m) -> int {
        const int D = 2 * m + (b & 1), j = b >> 1;
        return 32 * (D >> 2) + 8 * (2 * (D & 1) + (j >> 2)) + 4 * ((D >> 1) & 1) + (j & 3);
    };

    // ds threads: one wave per local head, lane = 16 m + sample (lane groups m >= M_ idle): the partial dots of a lane's
    // (sample, m) are contiguous over the lanes (conflict-free LDS reads), the sum over m is a lane-group reduction
    const int ds_s = lane & 15, ds_hh = w, ds_m = lg;
    const float invH = 1.0f / (float)H;
    // softmax weight and upstream weight gradient of a step, fetched one step ahead by inline-asm loads (see
    // ws_dma_rows_asm: an ordinary load here would serialise the LDS-DMA with the compute); retired by the step's
    // s_waitcnt vmcnt(0)
    auto load_stats = [&](int64_t o0, float& pmv, float& dwb) {
        int64_t b = o0 + ds_s;
        b = b < o_end ? b : o_end - 1;
        const float* pp_ = p.probs + (b * H + hbase + ds_hh) * M_ + (ds_m < M_ ? ds_m : 0);
        const float* dw_ = p.d_attn_w ? p.d_attn_w + b * M_ + (ds_m < M_ ? ds_m : 0) : pp_;
        if (w < HBL) {                                             // wave-uniform
            asm volatile("global_load_dword %0, %1, off" : "=v"(pmv) : "v"(pp_) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(dwb) : "v"(dw_) : "memory");
        }
    };

    f32x4 uacc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) uacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float pmv = 0.f, dwb = 0.f, pmv_n = 0.f, dwb_n = 0.f;
    // the weights are USED here, so that hipcc retires their loads now: left to the first MFMA it would put counted
    // s_waitcnt vmcnt(N) inside the loop, which (the inline-asm copies are invisible to its count) would wait for the
    // copy issued at the top of every step
#pragma unroll
    for (int ks = 0; ks < KJ; ++ks)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) asm volatile("" : "+v"(wreg[ks][ct]));
    load_stats(o_beg, pmv, dwb);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int cur = 0;
    for (int64_t o0 = o_beg; o0 < o_end; o0 += 16, cur ^= 1) {
        __builtin_amdgcn_s_barrier();                              // this step's tiles visible; the other buffers free
        if (o0 + 16 < o_end) {
            issue(o0 + 16, cur ^ 1);
            load_stats(o0 + 16, pmv_n, dwb_n);
        }
        const char* tb = db + cur * DT;
        // heads in groups of HG: P_h by MFMA, then the dot with this lane's x values (8-byte LDS reads at a per-tile base +
        // an immediate; one read and one unpack serve the HG heads of the group).  Every lane writes its partial (16 NCT
        // columns of one sample) to part[wave, lane group][head, m][sample]; the ds threads add the 32 of them.
        constexpr int HG = (HBL % 2 == 0 && !PKDOT) ? 2 : 1;
        int xa[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) xa[ct] = cur * XT + (WIDE ? (xaddr0 ^ (64 * (ct >> 1))) : (xaddr0 ^ (32 * ct)));
        float* pw = part + w * PSTR + r16;
#pragma unroll
        for (int h0 = 0; h0 < HBL; h0 += HG) {
            f32x4 P[HG][NCT];
#pragma unroll
            for (int g = 0; g < HG; ++g) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) P[g][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kl = 0; kl < HK; ++kl) {
                    const int ks = (h0 + g) * HK + kl;
                    const u32x4 bf = *reinterpret_cast<const u32x4*>(tb + daddr[ks & 3] + (ks >> 2) * 256);
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) P[g][ct] = X::mma(wreg[ks][ct], bf, P[g][ct]);
                    if (HILO) {
                        const u32x4 bl = *reinterpret_cast<const u32x4*>(dbl + cur * DT + daddr[ks & 3] + (ks >> 2) * 256);
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) P[g][ct] = X::mma(wreg[ks][ct], bl, P[g][ct]);
                    }
                }
            }
            unsigned int pk[HG][NCT][2];
            if (PKDOT) {
#pragma unroll
                for (int g = 0; g < HG; ++g)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        pk[g][ct][0] = pack_bf16x2(P[g][ct][0], P[g][ct][1]);
                        pk[g][ct][1] = pack_bf16x2(P[g][ct][2], P[g][ct][3]);
                    }
            }
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                float a[HG];
#pragma unroll
                for (int g = 0; g < HG; ++g) a[g] = 0.f;
                if (WIDE) {
#pragma unroll
                    for (int c2 = 0; c2 < NCT / 2; ++c2) {
                        const u32x4 xv = *reinterpret_cast<const u32x4*>(xb + xa[2 * c2] + m * ROWX);
                        float xf[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            xf[2 * i] = __uint_as_float(xv[i] << 16);
                            xf[2 * i + 1] = __uint_as_float(xv[i] & 0xffff0000u);
                        }
#pragma unroll
                        for (int g = 0; g < HG; ++g)
#pragma unroll
                            for (int i = 0; i < 8; ++i) a[g] = fmaf(P[g][2 * c2 + (i >> 2)][i & 3], xf[i], a[g]);
                    }
                } else
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const u32x2 xv = *reinterpret_cast<const u32x2*>(xb + xa[ct] + m * ROWX);
                    if (PKDOT) {
#pragma unroll
                        for (int g = 0; g < HG; ++g) {
                            a[g] = dot2_bf16(pk[g][ct][0], xv[0], a[g]);
                            a[g] = dot2_bf16(pk[g][ct][1], xv[1], a[g]);
                        }
                    } else {
                        const float x0 = __uint_as_float(xv[0] << 16), x1 = __uint_as_float(xv[0] & 0xffff0000u);
                        const float x2 = __uint_as_float(xv[1] << 16), x3 = __uint_as_float(xv[1] & 0xffff0000u);
#pragma unroll
                        for (int g = 0; g < HG; ++g) {
                            a[g] = fmaf(P[g][ct][0], x0, a[g]);
                            a[g] = fmaf(P[g][ct][1], x1, a[g]);
                            a[g] = fmaf(P[g][ct][2], x2, a[g]);
                            a[g] = fmaf(P[g][ct][3], x3, a[g]);
                        }
                    }
                }
                if (NOBR) {
                    // the same sums as a reduce-scatter: one row swap pairs the group's two heads (even rows end with head g = 0,
                    // odd rows with g = 1, each summed over a row pair), one half swap adds the halves; every lane then holds
                    // the total of head h0 + (lg & 1) and stores it -- lanes 32 apart store the same value to the same word.
                    // No exec-mask branch per value (12 per step before): the step's dot phase is ONE basic block, so the
                    // next modality's x reads can be scheduled above this one's arithmetic.
                    const unsigned int u0 = __float_as_uint(a[0]), u1 = __float_as_uint(a[HG - 1]);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(u0, u1, false, false);
                    const unsigned int us = __float_as_uint(__uint_as_float(r1[0]) + __uint_as_float(r1[1]));
                    const auto r2 = __builtin_amdgcn_permlane32_swap(us, us, false, false);
                    pw[((h0 + (HG == 2 ? (lg & 1) : 0)) * M_ + m) * 16] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
                } else
#pragma unroll
                for (int g = 0; g < HG; ++g) {
                    // sum over the wave's four lane groups (lanes r16, r16 + 16, + 32, + 48): rows swapped pairwise, then halves
                    const unsigned int ua = __float_as_uint(a[g]);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(ua, ua, false, false);
                    const float s1 = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
                    const unsigned int us = __float_as_uint(s1);
                    const auto r2 = __builtin_amdgcn_permlane32_swap(us, us, false, false);
                    const float s2 = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
                    if (lg == 0) pw[((h0 + g) * M_ + m) * 16] = s2;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" : "+v"(pmv), "+v"(dwb));
        if (w < HBL) {                               // whole waves: the lane-group sum below needs every lane
            float da = 0.f;
            if (ds_m < M_) {
                const float* pr = part + (ds_hh * M_ + ds_m) * 16 + ds_s;
#pragma unroll
                for (int i = 0; i < NPART; ++i) da += pr[i * PSTR];                      // fixed order
            }
            float dw = p.d_attn_w ? dwb : 0.f;
            if (p.d_entropy && ds_m < M_) {                        // eval mode: the entropy keeps its graph (ref :150-156)
                int64_t be = o0 + ds_s;
                be = be < o_end ? be : o_end - 1;
                float hsum = 0.f;
#pragma unroll
                for (int m = 0; m < M_; ++m) hsum -= xlogx(p.attn_w[be * M_ + m]);
                const bool live = (hsum >= 0.f) && (hsum <= p.log_M);
                dw += live ? -(logf(p.attn_w[be * M_ + ds_m]) + 1.0f) * p.d_entropy[be] : 0.f;
            }
            const float dp = da + dw * invH;
            const float dot = reduce_lg((ds_m < M_) ? pmv * dp : 0.f);     // over the M modalities of (sample, head)
            const int64_t b = o0 + ds_s;
            if (ds_m < M_) {
                const float d = (b < o_end) ? pmv * (dp - dot) : 0.f;
                if (b < o_end) p.dsbuf[(b * H + hbase + ds_hh) * M_ + ds_m] = d;
                const unsigned short hi = X::from_f32(d);
                dsh[ds_hh * DSROW + kslot(ds_s, ds_m)] = hi;
                dsl[ds_hh * DSROW + kslot(ds_s, ds_m)] = X::from_f32(d - X::to_f32(hi));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // u^T[k, h] += x^T[k, (b,m)] ds[(b,m), h]:  A operand by transposed reads of the x tile (rows = K index)
#pragma unroll
        for (int ks = 0; ks < KU; ++ks) {
            const u32x4 bh = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(dsh) + (r16 * DSROW + 32 * ks + 8 * lg) * 2);
            const u32x4 bl = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(dsl) + (r16 * DSROW + 32 * ks + 8 * lg) * 2);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const u32x4 af = tr_frag16(smem, (uz[ks][0] ? 0 : cur * XT) + (ua[ks][0] ^ (32 * ct)),
                                           (uz[ks][1] ? 0 : cur * XT) + (ua[ks][1] ^ (32 * ct)));
                uacc[ct] = X::mma(af, bh, uacc[ct]);
                uacc[ct] = X::mma(af, bl, uacc[ct]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // next tiles and statistics landed (a whole step to do so)
        asm volatile("" : "+v"(pmv_n), "+v"(dwb_n));
        pmv = pmv_n; dwb = dwb_n;
    }
    // u slab of this row chunk: rows = the block's heads, lane (lg, r16 = local head) holds k = ncol0 + 16 ct + 4 lg .. + 3
    if (r16 < HBL) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
            *reinterpret_cast<f32x4*>(u_slab + ((int64_t)chunk_u * H + hbase + r16) * E + ncol0 + 16 * ct + 4 * lg) = uacc[ct];
    }
}

// (A second form of this kernel around WAVE-PRIVATE x slabs -- dsu_slab_kernel, round 4: one barrier per step instead of three,
// every wave runs the softmax backward for all heads -- measured level with it, 118 against 113-120 us,
// profiles/r04_c2_dsu_slab_ablation.txt; it left the product library in round 5: tools/micro/variants/aecf_gemm_ws_ablations.hip.)

template <int KT, int KJ, int HK, int M_>
int launch_dsu_t(const BwdGArgs& a, float* u_slab, hipStream_t s) {
    constexpr int E = 32 * KT, JB = 32 * KJ, HBL = KJ / HK;
    constexpr int XROWS = 16 * M_, ZROWS = (32 * ((XROWS + 31) / 32) - XROWS) > 0 ? 1 : 0;
    const size_t smem_base = (size_t)(2 * XROWS + ZROWS) * 2 * E + (size_t)2 * 16 * 2 * JB + (size_t)8 * (16 * HBL * M_) * 4 +
                             (size_t)2 * 16 * 72 * 2;
    const bool hilo = a.do_lo && smem_base + (size_t)2 * 16 * 2 * JB <= 160 * 1024;      // (else: the default key-side accuracy)
    const size_t smem = smem_base + (hilo ? (size_t)2 * 16 * 2 * JB : 0);
    const int groups = E / JB;
    int64_t chunks = 256 / groups;
    if (chunks < 1) chunks = 1;
    int64_t rpb = (a.B + chunks - 1) / chunks;
    rpb = (rpb + 15) / 16 * 16;
    const int64_t nchunk = (a.B + rpb - 1) / rpb;
    dim3 grid(xcd_grid((unsigned)nchunk, (unsigned)groups)), block(512);
    // dot phase: 16-byte x reads and branch-free partial dots (VAR 2) where the K-step count allows, else the 8-byte form (VAR 0)
    constexpr int VAR = (KT / 4) % 2 == 0 ? 2 : 0;
    if (hilo) {                                                   // AECF_HILO_GRADS: P from do_hi + do_lo
        auto kern = dsu_ws_kernel<KT, KJ, HK, M_, false, VAR, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        kern<<<grid, block, smem, s>>>(a, u_slab, (int)rpb, (int)nchunk);
        return (int)nchunk;
    }
    auto kern = dsu_ws_kernel<KT, KJ, HK, M_, false, VAR>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a, u_slab, (int)rpb, (int)nchunk);
    return (int)nchunk;
}

template <int KT, int HK, int M_>
void launch_dx2(const BwdGArgs& a, hipStream_t s) {
    constexpr int K = 32 * KT;
    constexpr int CT = KT <= 16 ? 2 : 1;
    constexpr int KX = (3 * (KT / HK) + 31) / 32;
    const size_t smem = (size_t)2 * 16 * 2 * K + (size_t)2 * 16 * (((KT / HK) * M_) | 1) * sizeof(float) + (size_t)2 * 16 * M_ * (64 * KX + 16);
    const int groups = a.E / (128 * CT);
    int64_t chunks = (a.cu_budget > 0 ? a.cu_budget : 256) / groups;
    if (chunks < 1) chunks = 1;
    int64_t rpb = (a.B + chunks - 1) / chunks;
    rpb = (rpb + 15) / 16 * 16;
    const int64_t nchunk = (a.B + rpb - 1) / rpb;
    dim3 grid(xcd_grid((unsigned)nchunk, (unsigned)groups)), block(512);
    auto kern = dx_ws2_kernel<KT, HK, M_, CT>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a, (int)rpb, (int)nchunk);
}

template <int KT, int HK>
void launch_dx2_m(const BwdGArgs& a, hipStream_t s) {
    switch (a.M) {
        case 1: launch_dx2<KT, HK, 1>(a, s); break;
        case 2: launch_dx2<KT, HK, 2>(a, s); break;
        case 3: launch_dx2<KT, HK, 3>(a, s); break;
        default: launch_dx2<KT, HK, 4>(a, s); break;
    }
}

// heads of HK K-steps (head_dim = 32 HK); only the (K, head_dim) pairs with a power-of-two or small head count are built
template <int KT>
bool launch_dx2_hk(const BwdGArgs& a, hipStream_t s) {
    const int hk = a.hd / 32;
    if (hk * a.H != KT) return false;
    if (hk == 1 && KT <= 16) { launch_dx2_m<KT, 1>(a, s); return true; }          // up to 16 heads of 32
    if (hk == 2 && KT % 2 == 0 && KT <= 32) { launch_dx2_m<KT, (KT % 2 == 0 ? 2 : 1)>(a, s); return true; }
    if (hk == 3 && KT % 3 == 0) { launch_dx2_m<KT, (KT % 3 == 0 ? 3 : 1)>(a, s); return true; }
    if (hk == 4 && KT % 4 == 0) { launch_dx2_m<KT, (KT % 4 == 0 ? 4 : 1)>(a, s); return true; }
    if (hk == 8 && KT % 8 == 0) { launch_dx2_m<KT, (KT % 8 == 0 ? 8 : 1)>(a, s); return true; }
    if (hk == KT) { launch_dx2_m<KT, KT>(a, s); return true; }                     // a single head
    return false;
}


// ------------------------------------------------------------------------------------------------------------------
// Gated value projection, K = 512, with the tile stored as PER-WAVE COLUMN SLABS (the hot shape's form).
// In gemm_ws_kernel<VPROJ, GATE> a step has two barriers: every wave multiplies a K-slice of the tile against the folded
// key matrix, the partial scores meet in LDS, barrier, softmax -- and only then the value MFMAs start; all eight waves
// sit through the score round trip.  Here wave w COPIES exactly the K-slice it scores: bytes [128 w, 128 w + 128) of every
// (sample, modality) row go to its own slab of the LDS tile (six 1 KB LDS-DMA pieces of 8 rows x 128 B), so the moment its
// own copy of tile s + 1 has landed (the end-of-step vmcnt wait it does anyway) it can form its partial scores of step
// s + 1 -- no other wave's data is involved -- and park them in the other of two partial buffers.  The step's one barrier then
// publishes tile and partials together, the softmax follows it directly and the MFMA phase starts without a second barrier.
// LDS image of a slab: rows in modality-major order (row = 16 m + sample), 128 bytes each, 16-byte chunk c at
// c ^ (sample & 7): the fragment reads of both uses (16 samples of one modality, chunk 4 (ks & 1) + lane group) are
// bank-conflict free for ds_read_b128's lane grouping.
// Measured (C2, same-box A/B, 3 pairs): value projection 118 -> 114 us, step -0.6 %: the second barrier was a small part
// of the score phase; its MFMAs, the partial sums' trip through LDS and the softmax remain.
template <int M_>
__global__ __launch_bounds__(512, 2) void vproj_slab_kernel(GemmNtArgs p, int rows_per_block, int nchunk) {
    using X = Tr<BF16>;
    constexpr int KT = 16, K = 512, CT = 2, KG = 2;
    constexpr int SB = 128;                                        // slab row bytes (two K-steps)
    constexpr int SLAB = 16 * M_ * SB;                             // one wave's slab of a tile
    constexpr int TILE = 8 * SLAB;                                 // = 16 M rows x 1 KB
    constexpr int CW = 32, BC = 256, NV = 8;
    constexpr int GP = 8 * M_ * 256;                               // floats of one partial-score buffer
    constexpr int NDMA = 16 * M_ * SB / 1024;                      // 1 KB pieces per wave and tile (M = 3: 6)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gpart = reinterpret_cast<float*>(smem + 2 * TILE);      // [2][8 waves][M][16 heads][16 samples]

    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int N = p.N, H = p.H;
    unsigned int chunk_u, group_u;
    if (!xcd_tile(blockIdx.x, (unsigned)nchunk, (unsigned)(p.N / BC), chunk_u, group_u)) return;
    const int ncol0 = (int)group_u * BC + CW * w;
    const int64_t o_beg = (int64_t)chunk_u * rows_per_block;
    const int64_t o_end = (o_beg + rows_per_block) < p.R ? (o_beg + rows_per_block) : p.R;
    if (o_beg >= o_end) return;

    const char* asrc = reinterpret_cast<const char*>(p.a);
    // this wave's copy of a tile: piece i, lane l -> slab row 8 i + (l >> 3) = 16 m + sample, physical chunk l & 7
    const unsigned int row_pitch = (unsigned)K * 2u;               // bytes of one (sample, modality) row
    auto issue_piece = [&](int64_t o0, int buf, int i) {
        const int ov = (int)((o_end - o0) < 16 ? (o_end - o0) : 16);
        const char* src = asrc + o0 * M_ * (int64_t)row_pitch + SB * w;
        const int rho = 8 * i + (lane >> 3), m = rho >> 4, b = rho & 15;
        const int bc = b < ov ? b : ov - 1;
        const unsigned int voff = (unsigned)(bc * M_ + m) * row_pitch + (unsigned)(((lane & 7) ^ (b & 7)) << 4);
        const unsigned int dst = (unsigned)(size_t)(lds_void_t*)(smem + buf * TILE + w * SLAB + 1024 * i);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is named as a clobber on purpose
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst)
                     : "memory", "m0");
#pragma clang diagnostic pop
    };
    auto issue = [&](int64_t o0, int buf) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) issue_piece(o0, buf, i);
    };
    issue(o_beg, 0);

    const unsigned short* wsrc = reinterpret_cast<const unsigned short*>(p.w);
    u32x4 wreg[KT][CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = ncol0 + 8 * (r16 >> 2) + 4 * c + (r16 & 3);
        const unsigned short* wr = wsrc + (int64_t)n * K + 8 * lg;
        const u32x4* wf = reinterpret_cast<const u32x4*>(p.w_frag) + ((int64_t)((ncol0 / CW) * CT + c) * KT) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) wreg[ks][c] = p.w_frag ? wf[ks * 64] : *reinterpret_cast<const u32x4*>(wr + 32 * ks);
    }
    float bias[NV];
    load_bias<NV>(reinterpret_cast<const unsigned short*>(p.bias), ncol0 + NV * lg, bias);
    const int head = ncol0 / p.hd;
    // operand read of (modality m, K-step ks) for this lane's sample: slab ks >> 1, row 16 m + r16, chunk 4 (ks & 1) + lg
    int xoff[KG];
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) xoff[kk] = r16 * SB + ((((4 * kk) + lg) ^ (r16 & 7)) << 4);

    u32x4 ga[KG][2];                                               // score operands (rows of A hi / lo = heads) of this wave's K-steps
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        const int ks = KG * w + kg;
        ga[kg][0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.g_ahi) + (int64_t)r16 * K + 32 * ks + 8 * lg);
        ga[kg][1] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.g_alo) + (int64_t)r16 * K + 32 * ks + 8 * lg);
    }
    unsigned int kp[M_], kp_next[M_];
    {   // the first step's mask bytes by ordinary loads (hipcc counts them)
        const int64_t b0 = (o_beg + r16) < o_end ? (o_beg + r16) : (o_end - 1);
#pragma unroll
        for (int m = 0; m < M_; ++m) {
            kp[m] = p.g_kpm ? (unsigned int)p.g_kpm[b0 * M_ + m] : 0u;
            kp_next[m] = 0u;
        }
    }
    auto load_kp = [&](int64_t o0) {
        if (p.g_kpm) {
            const int64_t b = (o0 + r16) < o_end ? (o0 + r16) : (o_end - 1);
            const uint8_t* kq = p.g_kpm + b * M_;
#pragma unroll
            for (int m = 0; m < M_; ++m) asm volatile("global_load_ubyte %0, %1, off" : "+v"(kp_next[m]) : "v"(kq + m) : "memory");
        }
    };
    // partial scores of the tile in `buf` from this wave's own slab -> partial buffer gp
    auto scores = [&](int buf, float* gp) {
        const char* sl = smem + buf * TILE + w * SLAB;
#pragma unroll
        for (int m = 0; m < M_; ++m) {
            f32x4 gacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const u32x4 xf = *reinterpret_cast<const u32x4*>(sl + m * 16 * SB + xoff[kg]);
                gacc = X::mma(ga[kg][0], xf, gacc);
                gacc = X::mma(ga[kg][1], xf, gacc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gp[((w * M_ + m) * 16 + 4 * lg + r) * 16 + r16] = gacc[r];   // head 4 lg + r, sample r16
        }
    };
    // the weights (and score operands) are used here so that hipcc retires their loads now, not inside the loop
#pragma unroll
    for (int ks = 0; ks < KT; ++ks)
#pragma unroll
        for (int c = 0; c < CT; ++c) asm volatile("" : "+v"(wreg[ks][c]));
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) asm volatile("" : "+v"(ga[kg][0]), "+v"(ga[kg][1]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // own slab of the first tile landed
    scores(0, gpart);

    int cur = 0;
    for (int64_t o0 = o_beg; o0 < o_end; o0 += 16, cur ^= 1) {
        const bool more = o0 + 16 < o_end;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's partial scores of this step are in LDS
        __builtin_amdgcn_s_barrier();                              // tile and partials of this step visible; other buffers free
        if (more) load_kp(o0 + 16);
        // ---- softmax over the modalities for (this wave's head, this lane's sample); partial sums in wave order
        float pm[M_];
        {
            const float* gp = gpart + cur * GP;
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                float a = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) a += gp[((ww * M_ + m) * 16 + head) * 16 + r16];
                if (kp[m] != 0u) a = -INFINITY;                    // torch functional.py:6554-6566
                pm[m] = a;
                mx = fmaxf(mx, a);
            }
            softmax_fast<M_>(pm, mx);
            const int64_t bq = o0 + r16;
            const bool writer = lg == 0 && ncol0 % p.hd == 0 && bq < o_end;
            if (writer) {                                          // ONE masked region (three before, one per modality)
                float* dst = const_cast<float*>(p.probs) + (bq * H + head) * M_;
#pragma unroll
                for (int m = 0; m < M_; ++m) dst[m] = pm[m];
            }
        }
        // ---- products: modality-major items, operand reads PF items ahead; the next tile's copy goes out among the first
        constexpr int NIT = M_ * KT, PF = VPROJ_PF;
        const char* tb = smem + cur * TILE;
        auto rd = [&](int i) -> u32x4 {
            const int ks = i % KT, m = i / KT;
            return *reinterpret_cast<const u32x4*>(tb + (ks >> 1) * SLAB + m * 16 * SB + xoff[ks & 1]);
        };
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float ov[NV];
        unsigned int vpk[M_][NV / 2];
#pragma unroll
        for (int j = 0; j < NV; ++j) ov[j] = 0.f;
        u32x4 xf[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; ++i) xf[i] = rd(i);
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (i + PF < NIT) xf[(i + PF) % (PF + 1)] = rd(i + PF);
            if (i == NIT / 8 && more) issue(o0 + 16, cur ^ 1);
            const int ks = i % KT;
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = X::mma(wreg[ks][c], xf[i % (PF + 1)], acc[c]);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
            if (ks == KT - 1) {
                const int m = i / KT;
                float v[NV];
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[4 * c + r] = acc[c][r] + bias[4 * c + r];
                        ov[4 * c + r] = fmaf(pm[m], v[4 * c + r], ov[4 * c + r]);
                    }
#pragma unroll
                for (int j = 0; j < NV / 2; ++j) vpk[m][j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's slab of the next tile (and its mask bytes) landed
#pragma unroll
        for (int m = 0; m < M_; ++m) { asm volatile("" : "+v"(kp_next[m])); kp[m] = p.g_kpm ? kp_next[m] : 0u; }
        if (more) scores(cur ^ 1, gpart + (cur ^ 1) * GP);         // next step's partial scores, from this wave's own slab
        const int64_t b = o0 + r16;
        if (b < o_end) {
            if (p.v_out) {
#pragma unroll
                for (int m = 0; m < M_; ++m)
                    store_packed<CT>(reinterpret_cast<unsigned short*>(p.v_out) + (b * M_ + m) * N + ncol0 + NV * lg, vpk[m]);
            }
            if (p.out_f32) store_cols_f32<CT>(reinterpret_cast<float*>(p.c) + b * N + ncol0 + NV * lg, ov);
            else store_cols<CT>(reinterpret_cast<unsigned short*>(p.c) + b * N + ncol0 + NV * lg, ov);
            if (p.c_lo) store_cols_lo<CT>(reinterpret_cast<unsigned short*>(p.c_lo) + b * N + ncol0 + NV * lg, ov);
        }
    }
}

template <int M_>
void launch_vproj_slab(const GemmNtArgs& a, hipStream_t s) {
    const size_t smem = (size_t)2 * 16 * M_ * 1024 + (size_t)2 * 8 * M_ * 256 * sizeof(float);
    const int groups = a.N / 256;
    int64_t chunks = 256 / groups;
    if (chunks < 1) chunks = 1;
    int64_t rpb = (a.R + chunks - 1) / chunks;
    rpb = (rpb + 15) / 16 * 16;
    const int64_t nchunk = (a.R + rpb - 1) / rpb;
    dim3 grid(xcd_grid((unsigned)nchunk, (unsigned)groups)), block(512);
    auto kern = vproj_slab_kernel<M_>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a, (int)rpb, (int)nchunk);
}

template <int KT, int MODE, int M_, bool GATE>
void launch_ws(const GemmNtArgs& a, hipStream_t s) {
    constexpr int K = 32 * KT;
    constexpr int CT = KT <= 16 ? 2 : 1;                           // 32 columns per wave up to K = 512, 16 beyond
    constexpr int RT = MODE != WS_VPROJ ? 2 : M_;
    constexpr int OROWS = MODE != WS_VPROJ ? 32 : 16;
    size_t smem = (size_t)ws_plain_bufs(MODE, KT) * 16 * RT * 2 * K;
    const int groups = a.N / (128 * CT);
    // about one block per CU (256): chunks of whole steps
    int64_t chunks = 256 / groups;
    if (chunks < 1) chunks = 1;
    int64_t rpb = (a.R + chunks - 1) / chunks;
    rpb = (rpb + OROWS - 1) / OROWS * OROWS;
    const int64_t nchunk = (a.R + rpb - 1) / rpb;
    dim3 grid(xcd_grid((unsigned)nchunk, (unsigned)groups)), block(512);
    if (GATE) smem += (size_t)8 * M_ * 256 * sizeof(float);
    if (MODE == WS_VPROJ && GATE && KT == 16 && M_ <= 3) {         // hot shape (K = 512): the column-slab form
        if (!env_no_slab()) { launch_vproj_slab<(M_ <= 3 ? M_ : 3)>(a, s); return; }
    }
    auto kern = gemm_ws_kernel<KT, MODE, M_, GATE, CT>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a, (int)rpb, (int)nchunk);
}

template <int MODE, int M_, bool GATE>
void launch_kt(const GemmNtArgs& a, hipStream_t s) {
    switch (a.K / 32) {
        case 4: launch_ws<4, MODE, M_, GATE>(a, s); break;
        case 8: launch_ws<8, MODE, M_, GATE>(a, s); break;
        case 12: launch_ws<12, MODE, M_, GATE>(a, s); break;
        case 16: launch_ws<16, MODE, M_, GATE>(a, s); break;
        case 24: launch_ws<24, MODE, M_, GATE>(a, s); break;
        default: launch_ws<32, MODE, M_, GATE>(a, s); break;
    }
}

}  // namespace

// shapes the weight-stationary kernel takes (bf16 only); everything else stays on the tiled kernels
static bool ws_k_ok(int K) { return K == 128 || K == 256 || K == 384 || K == 512 || K == 768 || K == 1024; }

// value projection with M = 4 whose per-sample tile (16 samples x 4 rows x K, twice) does not fit LDS: flat-row form
static bool ws_vproj_flat(const GemmNtArgs& a) {
    const size_t tile = (size_t)16 * a.M * 2 * a.K;
    return (a.pooled & 1) && a.M == 4 && !a.g_ahi && 2 * tile > 150 * 1024 && (size_t)2 * 32 * 2 * a.K <= 150 * 1024;
}

bool gemm_ws_supported(const GemmNtArgs& a) {
    if (!ws_k_ok(a.K)) return false;
    if (a.out_f32 && (a.pooled & 1) && ws_vproj_flat(a)) return false;    // (the flat-row form stores bf16 only)
    const int ct = a.K <= 512 ? 2 : 1;
    if (a.N % (128 * ct) != 0) return false;
    const bool vproj = (a.pooled & 1) != 0;
    if (vproj && ws_vproj_flat(a)) return a.hd % (16 * ct) == 0 && a.lda == (int64_t)a.M * a.K;
    const size_t tile = (size_t)16 * (vproj ? a.M : 2) * 2 * a.K;          // LDS: two tiles (+ the score partials)
    size_t smem = 2 * tile + (vproj && a.g_ahi ? (size_t)8 * a.M * 1024 : 0);
    if (smem > 150 * 1024) return false;
    if (vproj) return a.M >= 1 && a.M <= 4 && a.hd % 32 == 0 && a.lda == (int64_t)a.M * a.K;
    return a.lda == a.K;
}

void launch_gemm_ws(const GemmNtArgs& a, hipStream_t s) {
    if (ws_vproj_flat(a)) {
        GemmNtArgs f = a;                      // the (b, m) rows of x as B*M plain rows of K
        f.R = a.R * a.M;
        f.lda = a.K;
        launch_kt<WS_VFLAT, 4, false>(f, s);
        return;
    }
    if (a.pooled & 1) {
        switch (a.M) {
            case 1: if (a.g_ahi) launch_kt<WS_VPROJ, 1, true>(a, s); else launch_kt<WS_VPROJ, 1, false>(a, s); break;
            case 2: if (a.g_ahi) launch_kt<WS_VPROJ, 2, true>(a, s); else launch_kt<WS_VPROJ, 2, false>(a, s); break;
            case 3: if (a.g_ahi) launch_kt<WS_VPROJ, 3, true>(a, s); else launch_kt<WS_VPROJ, 3, false>(a, s); break;
            default: if (a.g_ahi) launch_kt<WS_VPROJ, 4, true>(a, s); else launch_kt<WS_VPROJ, 4, false>(a, s); break;
        }
    } else {
        launch_kt<WS_PLAIN, 1, false>(a, s);
    }
}

// score gradient + key-side batch reduction from x (no saved V); returns the number of u slabs written (row chunks), or 0
// when the shape is not taken (caller falls back to dscore_v / bwd_g + u_stream)
int dsu_ws_chunks(const BwdGArgs& a) {
    if (a.M < 1 || a.M > 4 || a.hd % 32 != 0 || a.E != a.H * a.hd) return 0;
    if (a.E != 256 && a.E != 512) return 0;
    const int jb = 256;
    if (jb % a.hd != 0) return 0;
    const int hk = a.hd / 32;
    if (hk != 1 && hk != 2 && hk != 4 && hk != 8) return 0;
    {   // LDS: two x tiles + K-padding page + two do tiles + partial dots + ds operand arrays
        const int xrows = 16 * a.M, zrows = (32 * ((xrows + 31) / 32) - xrows) > 0 ? 1 : 0, hbl = jb / a.hd;
        const size_t smem = (size_t)(2 * xrows + zrows) * 2 * a.E + (size_t)2 * 16 * 2 * jb + (size_t)8 * (16 * hbl * a.M) * 4 + 4608;
        if (smem > 160 * 1024) return 0;
    }
    const int groups = a.E / jb;
    int64_t chunks = 256 / groups;
    int64_t rpb = (a.B + chunks - 1) / chunks;
    rpb = (rpb + 15) / 16 * 16;
    return (int)((a.B + rpb - 1) / rpb);
}

template <int KT, int HK>
static int launch_dsu_m(const BwdGArgs& a, float* u_slab, hipStream_t s) {
    switch (a.M) {
        case 1: return launch_dsu_t<KT, 8, HK, 1>(a, u_slab, s);
        case 2: return launch_dsu_t<KT, 8, HK, 2>(a, u_slab, s);
        case 3: return launch_dsu_t<KT, 8, HK, 3>(a, u_slab, s);
        default: return launch_dsu_t<KT, 8, HK, 4>(a, u_slab, s);
    }
}

int launch_dsu_ws(const BwdGArgs& a, float* u_slab, hipStream_t s) {
    if (dsu_ws_chunks(a) == 0) return 0;
    const int hk = a.hd / 32;
#define DSU_CASE(KT_)                                                   \
    switch (hk) {                                                       \
        case 1: return launch_dsu_m<KT_, 1>(a, u_slab, s);              \
        case 2: return launch_dsu_m<KT_, 2>(a, u_slab, s);              \
        case 4: return launch_dsu_m<KT_, 4>(a, u_slab, s);              \
        default: return launch_dsu_m<KT_, 8>(a, u_slab, s);             \
    }
    if (a.E == 256) { DSU_CASE(8) }
    DSU_CASE(16)
#undef DSU_CASE
}

// dx through the weight-stationary engine (bf16); false = shape not taken (caller uses launch_bwd_g(dx = true))
bool launch_dx_ws(const BwdGArgs& a, hipStream_t s) {
    if (env_no_ws()) return false;
    if (a.M < 1 || a.M > 4) return false;
    if (a.hd % 32 != 0 || a.E != a.H * a.hd || 16 * a.H * a.M > 1024) return false;
    switch (a.E) {
        case 256: return launch_dx2_hk<8>(a, s);
        case 512: return launch_dx2_hk<16>(a, s);
        case 768: return launch_dx2_hk<24>(a, s);
        case 1024: return launch_dx2_hk<32>(a, s);
        default: return false;
    }
}

}  // namespace aecf

