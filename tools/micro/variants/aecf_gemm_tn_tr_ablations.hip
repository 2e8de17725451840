// gemm_tn, bf16, gfx950 transposed-LDS-read form:  out[j][k] = sum_b lhs[b][j] * rhs(b,k)   (see aecf_gemm_tn.hip).
//
// The reduction index (the batch) is the slow axis of both row-major operands.  gfx950 can read an MFMA operand
// TRANSPOSED out of LDS (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block of 16-bit values and
// every lane receives one column), so the tiles stay in their natural [batch row][feature] order:
//   * lhs tile  [64 batch rows][128 features]: global_load_lds (LDS-DMA), double buffered -- no register round
//     trip, no vector instructions at all;
//   * rhs tile, plain GEMM: the same;
//   * rhs tile, POOLED: rhs(b,k) = sum_m probs[b, head(j), m] * x[b,m,k].  Each thread loads two 16-byte chunks of
//     each modality row, forms the pooled chunk per head slot (v_pk_fma_f32 against (p, p) pairs, one v_cvt_pk_bf16_f32 per pair) and
//     writes it row-major with one ds_write_b128 -- no transposition work;
//   * lhs column sums (the bias gradients): one extra MFMA against an all-ones operand.
// LDS image of a [64][128] tile: 8-row x 32-column subtiles of 512 B, 16-byte chunk ch of row r at
//     2048 (r >> 3) + 512 (ch >> 2) + 64 (r & 7) + 16 ((ch & 3) ^ ((r >> 2) & 3))
// (cdna_hip_programming.md T10, image (a)): the DMA writes, the b128 writes and the transposed reads are bank-conflict
// free, and all 24 transposed reads of a wave's step share 4 address registers (the rest are immediates).
// Block = 512 threads (8 waves as 4 (j) x 2 (k)), block tile 128 x 128, wave tile 32 x 64, 64 batch rows per step;
// the pooled product also has a 1024-thread form with a 256 x 128 tile (gemm_tn_tr_wide_kernel below), used when it fits.
// Output: float32 partial slabs per batch split (deterministic; reduced by reduce_segments).
#include <stdlib.h>
#include <type_traits>

#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

namespace {

constexpr int TRB = 64;                        // batch rows per step
constexpr int TR_TILE = TRB * 256;             // bytes of one [64][128 bf16] tile

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;

__device__ __forceinline__ int tr_off(int row, int ch) {
    return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// LDS-DMA of a [64 rows][256 B] tile.  The DMA destination is lane-linear: chunk c = tid + 512 i lands at byte 16 c,
// which in this image is row 8 (c >> 7) + ((c >> 2) & 7), chunk 4 ((c >> 5) & 3) + ((c & 3) ^ ((row >> 2) & 3)): the
// permutation goes on the per-lane SOURCE address (a wave-instruction still reads 8 rows x 128 contiguous bytes).
// Rows >= rows_valid re-read the last valid row, chunks >= chunks_valid re-read chunk 0 (in bounds; the caller
// zeroes or never stores what they produce).  src is wave-uniform; the per-lane part is a 32-bit offset.
// The copy is issued through inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every transposed LDS read that
// follows a global_load_lds it can see (the ds_read_tr builtin carries no alias information), which would expose the
// whole memory latency each step.  The caller retires the copy with its own s_waitcnt vmcnt(0) one step later.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is named as a clobber on purpose
__device__ __forceinline__ void dma_tile_tr_async(const char* __restrict__ src, unsigned int ld_bytes, int rows_valid,
                                                  int chunks_valid, char* lds) {
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = threadIdx.x + 512 * i;
        const int row = 8 * (c >> 7) + ((c >> 2) & 7);
        const int rowc = row < rows_valid ? row : rows_valid - 1;
        int logical = 4 * ((c >> 5) & 3) + ((c & 3) ^ ((row >> 2) & 3));
        logical = logical < chunks_valid ? logical : 0;
        const unsigned int voff = (unsigned)rowc * ld_bytes + (unsigned)logical * 16u;
        const unsigned int dst = (unsigned)(size_t)(lds_void_t*)(lds + (wbase + 512 * i) * 16);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(voff), "s"(src), "s"(dst) : "memory", "m0");
    }
}
#pragma clang diagnostic pop

// NOTE on the pooling arithmetic: the probabilities sit in LDS as (p, p) PAIRS and are used by plain v_pk_mul_f32 /
// v_pk_fma_f32 (both halves read from their own register).  The form hipcc's SLP vectoriser produced from scalar code
// -- one register broadcast to both halves through op_sel -- gave run-to-run different low halves on MI355X at full
// size (profiles/r01_pmc_notes.md); this file is built with -fno-slp-vectorize and the full-size determinism test
// (tests/test_pool_gpu_large.py) guards the pair form.
// MFMA operand (8 consecutive batch rows 32 ks + 8 lg .. + 7 of feature column col0 + r16) by two transposed reads
__device__ __forceinline__ u32x4 tr_frag(const char* tile, int addr_lo, int addr_hi) {
    const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(tile + addr_lo));
    const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(tile + addr_hi));
    const u32x2 l = __builtin_bit_cast(u32x2, lo), h = __builtin_bit_cast(u32x2, hi);
    return u32x4{l[0], l[1], h[0], h[1]};
}

template <int M_, bool POOLED, int MAXS>
__global__ __launch_bounds__(512, (M_ <= 3 ? 4 : 2)) void gemm_tn_tr_kernel(GemmTnArgs p) {
    using X = Tr<BF16>;
    constexpr int RT = 2, CT = 4;
    constexpr int PLN = POOLED ? (TRB * MAXS * M_ + 511) / 512 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int E = p.E, H = p.H;
    const int EJ = p.Ej > 0 ? p.Ej : p.E;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    if (POOLED && p.dq.w_k) dqp_rows<BF16>(p.dq, (int)blockIdx.x, (int)gridDim.x);   // side job: dq' for the finalize launch

    const unsigned int nK = (unsigned)((E + 127) / 128), nJt = (unsigned)((EJ + 127) / 128);
    unsigned int split_u, tile_u;
    if (!xcd_tile(blockIdx.x, (unsigned)p.splits, nK * nJt, split_u, tile_u)) return;
    const int kt_idx = (int)(tile_u % nK), jt_idx = (int)(tile_u / nK);
    const int j0 = jt_idx * 128, k0 = kt_idx * 128;
    const int split = (int)split_u;
    const int64_t rbeg = (int64_t)split * p.rows_per_split;
    const int64_t rend = (rbeg + p.rows_per_split) < p.B ? (rbeg + p.rows_per_split) : p.B;
    const int jrows = (EJ - j0) >= 128 ? 128 : (EJ - j0);     // multiples of 64
    const int kcols = (E - k0) >= 128 ? 128 : (E - k0);

    const int h_first = POOLED ? j0 / p.hd : 0;
    const int h_last = POOLED ? (j0 + jrows - 1) / p.hd : 0;
    const int nslots = h_last - h_first + 1;

    // LDS carve: lhs tile x2 | rhs tiles (POOLED: one per head slot; plain: x2) | probabilities x2
    char* ldsL = smem;
    char* ldsR = smem + 2 * TR_TILE;
    f32x2* pl = reinterpret_cast<f32x2*>(ldsR + (POOLED ? MAXS : 2) * TR_TILE);       // [TRB][MAXS][M] (p, p) pairs

    // wave tile 32 (j) x 64 (k)
    const int j0w = 32 * (w >> 1), k0w = 64 * (w & 1);
    const bool wave_on = j0w < jrows && k0w < kcols;
    const int wslot = POOLED ? ((j0 + (j0w < jrows ? j0w : 0)) / p.hd - h_first) : 0;
    const bool do_cs = p.colsum != nullptr && kt_idx == 0 && k0w == 0;

    // transposed-read addresses: lane 4q+pp of group lg reads row 32 ks + 8 lg + 4 hh + q, columns 4 pp .. 4 pp + 3
    // of the 16-column block blk (chunk 2 blk + (pp >> 1)):
    //   off = 8192 ks + 2048 lg + 512 (blk >> 1) + 256 hh + 64 q + 16 ((2 (blk & 1) + (pp >> 1)) ^ (2 (lg & 1) + hh)) + 8 (pp & 1)
    // -> 4 per-lane values tx[blk & 1][hh]; ks and blk >> 1 are immediates, the wave's block origin a uniform add.
    const int q = r16 >> 2, pp = r16 & 3;
    int tx[2][2];
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            tx[b1][hh] = 2048 * lg + 64 * q + 8 * (pp & 1) + 256 * hh + 16 * ((2 * b1 + (pp >> 1)) ^ (2 * (lg & 1) + hh));
    const int a_org = 512 * (w >> 1);                     // lhs block 2 (w >> 1) + rt
    const int b_org = 1024 * (w & 1);                     // rhs block 4 (w & 1) + ct

    f32x4 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csacc[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) csacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* lhs = reinterpret_cast<const char*>(p.lhs);
    const char* rhs = reinterpret_cast<const char*>(p.rhs);
    const unsigned int ldl = (unsigned)EJ * 2u;
    const unsigned int ldr = (unsigned)(POOLED ? M_ : 1) * (unsigned)E * 2u;

    // pooling role of this thread: batch row prow, 16-byte chunks pc and pc + 8 of the 128-feature slice
    const int prow = threadIdx.x >> 3, pc = threadIdx.x & 7;
    const unsigned int prow_off = (unsigned)prow * ((unsigned)(POOLED ? M_ : 1) * (unsigned)p.E * 2u);
    const unsigned int coff[2] = {8 * pc < kcols ? 16u * pc : 0u, 8 * (pc + 8) < kcols ? 16u * (pc + 8) : 0u};
    u32x4 Rb[POOLED ? M_ : 1][2];
    float plr[PLN];

    auto issue_dma = [&](int64_t base, int buf) {
        const int nvalid = (int)((rend - base) < TRB ? (rend - base) : TRB);
        dma_tile_tr_async(lhs + base * (int64_t)ldl + (int64_t)j0 * 2, ldl, nvalid, jrows >> 3, ldsL + buf * TR_TILE);
        if (!POOLED) dma_tile_tr_async(rhs + base * (int64_t)ldr + (int64_t)k0 * 2, ldr, nvalid, kcols >> 3, ldsR + buf * TR_TILE);
    };
    // POOLED operand fetches are inline-asm loads issued ONE STEP AHEAD (hipcc would otherwise wait vmcnt(0) for them
    // in front of the first transposed read, i.e. expose the whole memory latency every step); the step's own
    // s_waitcnt vmcnt(0) retires them together with the lhs DMA tile they belong to.
    auto load_probs = [&](int64_t base) {
        const int nvalid = (int)((rend - base) < TRB ? (rend - base) : TRB);
        const float* pu = p.probs + (base * H + h_first) * M_;                  // wave-uniform base, 32-bit lane offsets
#pragma unroll
        for (int i = 0; i < PLN; ++i) {
            const int idx = threadIdx.x + 512 * i;              // (t, s, m) with MAXS slots per row
            const int t = idx / (MAXS * M_), rem = idx - t * (MAXS * M_);
            const int sl = rem / M_, m = rem - sl * M_;
            const bool on = idx < TRB * MAXS * M_ && sl < nslots && t < nvalid;
            const float* src = pu + (on ? (unsigned)(t * H * M_ + sl * M_ + m) : 0u);
            asm volatile("global_load_dword %0, %1, off" : "=v"(plr[i]) : "v"(src) : "memory");
        }
    };
    auto probs_on = [&](int64_t base, int i) -> bool {
        const int nvalid = (int)((rend - base) < TRB ? (rend - base) : TRB);
        const int idx = threadIdx.x + 512 * i;
        const int t = idx / (MAXS * M_), rem = idx - t * (MAXS * M_);
        return idx < TRB * MAXS * M_ && rem / M_ < nslots && t < nvalid;
    };
    // rows past nvalid re-read the last valid row (their probabilities are 0 -> pooled rows of exact zeros); chunks past
    // kcols re-read chunk 0 (never stored)
    auto load_x = [&](int64_t base, int c) {
        const int nvalid = (int)((rend - base) < TRB ? (rend - base) : TRB);
        const char* xu = rhs + base * (int64_t)ldr + (int64_t)k0 * 2;
        const unsigned int last = (unsigned)(nvalid - 1) * ldr;
        const unsigned int roff = (prow_off < last ? prow_off : last) + coff[c];
#pragma unroll
        for (int m = 0; m < M_; ++m) {
            const char* xm = xu + (size_t)m * E * 2;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(Rb[m][c]) : "v"(roff), "s"(xm) : "memory");
        }
    };
    // plain product: the next step's copy is issued BETWEEN the step's two K-steps, not in front of its first MFMAs (the
    // step end to end, timed over whole fwd+bwd steps, is 1.7 % shorter that way; same-box A/B, 5 of 5)
    int64_t next_base = -1;
    auto mma_phase = [&](int cur, int nvalid_cur) {
        if (wave_on) {
            const char* lt = ldsL + cur * TR_TILE;
            const char* rt_tile = ldsR + (POOLED ? wslot : cur) * TR_TILE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (!POOLED && ks == 1 && next_base >= 0) issue_dma(next_base, cur ^ 1);
                u32x4 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    a[rt] = tr_frag(lt + a_org + 8192 * ks, tx[rt][0], tx[rt][1]);
                if (do_cs) {                              // row sums of lhs: product with ones over the rows that exist
                    u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                    if (nvalid_cur < TRB) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const int kk = 32 * ks + 8 * lg + 2 * d;
                            ones[d] = (kk < nvalid_cur ? 0x3f80u : 0u) | (kk + 1 < nvalid_cur ? 0x3f800000u : 0u);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) csacc[rt] = X::mma(a[rt], ones, csacc[rt]);
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const u32x4 b = tr_frag(rt_tile + b_org + 8192 * ks + 512 * (ct >> 1), tx[ct & 1][0], tx[ct & 1][1]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = X::mma(a[rt], b, acc[rt][ct]);
                }
            }
        }
    };

    if (POOLED) {
        if (rbeg < rend) {                                // (a trailing split can be empty: its slab is all zeros)
            issue_dma(rbeg, 0);
            load_probs(rbeg);
            load_x(rbeg, 0);
            load_x(rbeg, 1);
        }
        int cur = 0;
        for (int64_t base = rbeg; base < rend; base += TRB, cur ^= 1) {
            const bool more = base + TRB < rend;
            const int nvalid_cur = (int)((rend - base) < TRB ? (rend - base) : TRB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this step's lhs tile, x chunks and probabilities landed
#pragma unroll
            for (int i = 0; i < PLN; ++i) {
                asm volatile("" : "+v"(plr[i]));
                const int idx = threadIdx.x + 512 * i;
                const float pv = probs_on(base, i) ? plr[i] : 0.f;
                if (idx < TRB * MAXS * M_) pl[idx] = f32x2{pv, pv};
            }
#pragma unroll
            for (int m = 0; m < M_; ++m)
#pragma unroll
                for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(Rb[m][c]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // MFMAs of the previous step done; lhs tile + probabilities visible
            if (more) {
                issue_dma(base + TRB, cur ^ 1);
                load_probs(base + TRB);
            }
            const f32x2* plc = pl + prow * (MAXS * M_);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x2 xv[M_][4];
#pragma unroll
                for (int m = 0; m < M_; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        xv[m][i] = f32x2{__uint_as_float(Rb[m][c][i] << 16), __uint_as_float(Rb[m][c][i] & 0xffff0000u)};
#pragma unroll
                for (int m = 0; m < M_; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xv[m][i]));
                if (more) load_x(base + TRB, c);          // the chunk registers are free again: next step's chunk flies
                const int woff = tr_off(prow, pc + 8 * c);
                auto pool_slots = [&](auto lo_tag) {              // (two copies, one branch: see gemm_tn_tr_wide_kernel)
                    constexpr bool LO = decltype(lo_tag)::value;
#pragma unroll
                    for (int sl = 0; sl < MAXS; ++sl) {
                        if (sl < nslots) {
                            f32x2 pv[4];
                            const f32x2 p0 = plc[sl * M_];
#pragma unroll
                            for (int i = 0; i < 4; ++i) pv[i] = xv[0][i] * p0;
#pragma unroll
                            for (int m = 1; m < M_; ++m) {
                                const f32x2 pm = plc[sl * M_ + m];
#pragma unroll
                                for (int i = 0; i < 4; ++i) pv[i] = __builtin_elementwise_fma(xv[m][i], pm, pv[i]);
                            }
                            u32x4 o;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (LO) {                          // the low parts of the pooled values (AECF_HILO_GRADS)
                                    pv[i][0] -= X::to_f32(X::from_f32(pv[i][0]));
                                    pv[i][1] -= X::to_f32(X::from_f32(pv[i][1]));
                                }
                                o[i] = pack_bf16x2(pv[i][0], pv[i][1]);
                            }
                            *reinterpret_cast<u32x4*>(ldsR + sl * TR_TILE + woff) = o;
                        }
                    }
                };
                if (p.pool_lo) pool_slots(std::true_type()); else pool_slots(std::false_type());
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // pooled tiles visible
            mma_phase(cur, nvalid_cur);
        }
    } else {
        if (rbeg < rend) issue_dma(rbeg, 0);
        int cur = 0;
        for (int64_t base = rbeg; base < rend; base += TRB, cur ^= 1) {
            const bool more = base + TRB < rend;
            const int nvalid_cur = (int)((rend - base) < TRB ? (rend - base) : TRB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this step's tiles landed (issued one step ago)
            __builtin_amdgcn_s_barrier();                 // ... for every wave; MFMAs of the previous step done
            if (nvalid_cur < TRB) {                       // ragged last step: zero the rhs rows that do not exist
                for (int c = threadIdx.x; c < (TRB - nvalid_cur) * 16; c += 512)
                    *reinterpret_cast<u32x4*>(ldsR + cur * TR_TILE + tr_off(nvalid_cur + (c >> 4), c & 15)) = u32x4{0u, 0u, 0u, 0u};
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            next_base = more ? base + TRB : -1;
            if (more && !wave_on) issue_dma(base + TRB, cur ^ 1);     // (a wave without a tile still copies its share)
            mma_phase(cur, nvalid_cur);
        }
    }

    // ---- slab stores: acc[rt][ct][r] = out[j0 + j0w + 16 rt + 4 lg + r][k0 + k0w + 16 ct + r16] ----
    float* out = p.out + (int64_t)split * EJ * E;
    if (wave_on) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (k0w + 16 * ct < kcols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[(int64_t)(j0 + j0w + 16 * rt + 4 * lg + r) * E + k0 + k0w + 16 * ct + r16] = acc[rt][ct][r];
                }
        if (do_cs && r16 == 0) {                          // every column of the ones-product holds the row sums
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    p.colsum[(int64_t)split * EJ + j0 + j0w + 16 * rt + 4 * lg + r] = csacc[rt][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// WIDE pooled variant: 1024 threads = 16 waves as 8 (j) x 2 (k), block tile 256 (j) x 128 (k), ONE block per CU.
// The same pooled x slice now serves 4 head slots instead of 2: the bf16 unpack of x and the per-step address work are
// spent once per 256 output rows (half the vector instructions per output), x crosses L2 -> LDS twice instead of four
// times, and the wave count per CU is unchanged (16).  A thread pools ONE 16-byte chunk of a row (64 rows x 16 chunks).
// lhs tile = two 128-wide sub-tiles of the [64][128] image, double buffered (64 KB); rhs = MAXS pooled tiles.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma_subtile_1024(const char* __restrict__ src, unsigned int ld_bytes, int rows_valid,
                                                 int chunks_valid, char* lds) {
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    const int c = threadIdx.x;                                    // 1024 chunks = one [64][128] image
    const int row = 8 * (c >> 7) + ((c >> 2) & 7);
    const int rowc = row < rows_valid ? row : rows_valid - 1;
    int logical = 4 * ((c >> 5) & 3) + ((c & 3) ^ ((row >> 2) & 3));
    logical = logical < chunks_valid ? logical : 0;
    const unsigned int voff = (unsigned)rowc * ld_bytes + (unsigned)logical * 16u;
    const unsigned int dst = (unsigned)(size_t)(lds_void_t*)(lds + wbase * 16);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(src), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop

#ifdef AECF_TN_TIMELINE
// experiment build only (tools/debug/tn_timeline.py): shader-clock stamps of one wave at the phase boundaries of every step
__device__ unsigned long long g_tn_timeline[8 * 64 * 8];
#define TN_STAMP(slot) do { if (tl_on) { tl[(step_no * 8 + (slot))] = __builtin_readcyclecounter(); } } while (0)
#else
#define TN_STAMP(slot) do { } while (0)
#endif

template <int M_, int MAXS>
__global__ __launch_bounds__(1024, 4) void gemm_tn_tr_wide_kernel(GemmTnArgs p) {
    using X = Tr<BF16>;
    constexpr int RT = 2, CT = 4;
    constexpr int PLN = (TRB * MAXS * M_ + 1023) / 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int E = p.E, H = p.H;
    const int EJ = p.Ej > 0 ? p.Ej : p.E;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    if (p.dq.w_k) dqp_rows<BF16>(p.dq, (int)blockIdx.x, (int)gridDim.x);      // side job: dq' for the finalize launch (aecf_common.h)
#ifdef AECF_TN_TIMELINE
    // waves 0, 5, 10, 15 of blocks 0 and 300 (lane 0)
    const int tl_wsel = (w % 5 == 0) ? w / 5 : -1;
    const int tl_bsel = blockIdx.x == 0 ? 0 : (blockIdx.x == 300 ? 1 : -1);
    const bool tl_on = lane == 0 && tl_wsel >= 0 && tl_bsel >= 0;
    unsigned long long* tl = g_tn_timeline + (tl_bsel * 4 + (tl_wsel < 0 ? 0 : tl_wsel)) * 64 * 8;
    int step_no = 0;
#endif

    const unsigned int nK = (unsigned)((E + 127) / 128), nJt = (unsigned)((EJ + 255) / 256);
    unsigned int split_u, tile_u;
    if (!xcd_tile(blockIdx.x, (unsigned)p.splits, nK * nJt, split_u, tile_u)) return;
    const int kt_idx = (int)(tile_u % nK), jt_idx = (int)(tile_u / nK);
    const int j0 = jt_idx * 256, k0 = kt_idx * 128;
    const int split = (int)split_u;
    const int64_t rbeg = (int64_t)split * p.rows_per_split;
    const int64_t rend = (rbeg + p.rows_per_split) < p.B ? (rbeg + p.rows_per_split) : p.B;
    const int jrows = (EJ - j0) >= 256 ? 256 : (EJ - j0);     // multiples of 64
    const int kcols = (E - k0) >= 128 ? 128 : (E - k0);

    const int h_first = j0 / p.hd;
    const int h_last = (j0 + jrows - 1) / p.hd;
    const int nslots = h_last - h_first + 1;

    // LDS carve: lhs [2 buffers][2 sub-tiles] | MAXS pooled rhs tiles | probabilities
    char* ldsL = smem;
    char* ldsR = smem + 4 * TR_TILE;
    f32x2* pl = reinterpret_cast<f32x2*>(ldsR + MAXS * TR_TILE);       // [TRB][MAXS][M] (p, p) pairs

    // wave tile 32 (j) x 64 (k)
    const int j0w = 32 * (w >> 1), k0w = 64 * (w & 1);
    const bool wave_on = j0w < jrows && k0w < kcols;
    const int wslot = (j0 + (j0w < jrows ? j0w : 0)) / p.hd - h_first;
    const bool do_cs = p.colsum != nullptr && kt_idx == 0 && k0w == 0;

    const int q = r16 >> 2, pp = r16 & 3;
    int tx[2][2];
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            tx[b1][hh] = 2048 * lg + 64 * q + 8 * (pp & 1) + 256 * hh + 16 * ((2 * b1 + (pp >> 1)) ^ (2 * (lg & 1) + hh));
    const int a_org = (j0w >> 7) * TR_TILE + 512 * ((w >> 1) & 3);   // sub-tile, then lhs block 2 ((w >> 1) & 3) + rt
    const int b_org = 1024 * (w & 1);

    f32x4 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csacc[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) csacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* lhs = reinterpret_cast<const char*>(p.lhs);
    const char* rhs = reinterpret_cast<const char*>(p.rhs);
    const unsigned int ldl = (unsigned)EJ * 2u;
    const unsigned int ldr = (unsigned)M_ * (unsigned)E * 2u;

    // pooling role of this thread: batch row prow, 16-byte chunk pc of the 128-feature slice.  (Round 4: a mapping that gives
    // each 8-lane group of the pooled tile's ds_write_b128 two rows x four chunks -- the 128-byte bank window exactly once
    // instead of the same 64 bytes twice -- measured level, 93-97 us either way: the stores are not on the critical path.)
    const int prow = threadIdx.x >> 4, pc = threadIdx.x & 15;
    const unsigned int prow_off = (unsigned)prow * ldr;
    const unsigned int coff = 8 * pc < kcols ? 16u * pc : 0u;
    u32x4 Rb[M_];
    float plr[PLN];

    auto issue_dma = [&](int64_t base, int buf) {
        const int nvalid = (int)((rend - base) < TRB ? (rend - base) : TRB);
        const char* src = lhs + base * (int64_t)ldl + (int64_t)j0 * 2;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
            if (128 * sub < jrows)                        // (block-uniform)
                dma_subtile_1024(src + 256 * sub, ldl, nvalid, (jrows - 128 * sub) >= 128 ? 16 : (jrows - 128 * sub) >> 3,
                                 ldsL + (2 * buf + sub) * TR_TILE);
    };
    auto probs_on = [&](int64_t base, int i) -> bool {
        const int nvalid = (int)((rend - base) < TRB ? (rend - base) : TRB);
        const int idx = threadIdx.x + 1024 * i;
        const int t = idx / (MAXS * M_), rem = idx - t * (MAXS * M_);
        return idx < TRB * MAXS * M_ && rem / M_ < nslots && t < nvalid;
    };
    auto load_probs = [&](int64_t base) {
        const float* pu = p.probs + (base * H + h_first) * M_;
#pragma unroll
        for (int i = 0; i < PLN; ++i) {
            const int idx = threadIdx.x + 1024 * i;             // (t, s, m) with MAXS slots per row
            const int t = idx / (MAXS * M_), rem = idx - t * (MAXS * M_);
            const int sl = rem / M_, m = rem - sl * M_;
            const float* src = pu + (probs_on(base, i) ? (unsigned)(t * H * M_ + sl * M_ + m) : 0u);
            asm volatile("global_load_dword %0, %1, off" : "=v"(plr[i]) : "v"(src) : "memory");
        }
    };
    auto load_x = [&](int64_t base) {
        const int nvalid = (int)((rend - base) < TRB ? (rend - base) : TRB);
        const char* xu = rhs + base * (int64_t)ldr + (int64_t)k0 * 2;
        const unsigned int last = (unsigned)(nvalid - 1) * ldr;
        const unsigned int roff = (prow_off < last ? prow_off : last) + coff;
#pragma unroll
        for (int m = 0; m < M_; ++m) {
            const char* xm = xu + (size_t)m * E * 2;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(Rb[m]) : "v"(roff), "s"(xm) : "memory");
        }
    };
    auto mma_phase = [&](int cur, int nvalid_cur) {
        if (wave_on) {
            const char* lt = ldsL + 2 * cur * TR_TILE;
            const char* rt_tile = ldsR + wslot * TR_TILE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    a[rt] = tr_frag(lt + a_org + 8192 * ks, tx[rt][0], tx[rt][1]);
                if (do_cs) {
                    u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                    if (nvalid_cur < TRB) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const int kk = 32 * ks + 8 * lg + 2 * d;
                            ones[d] = (kk < nvalid_cur ? 0x3f80u : 0u) | (kk + 1 < nvalid_cur ? 0x3f800000u : 0u);
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) csacc[rt] = X::mma(a[rt], ones, csacc[rt]);
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const u32x4 b = tr_frag(rt_tile + b_org + 8192 * ks + 512 * (ct >> 1), tx[ct & 1][0], tx[ct & 1][1]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = X::mma(a[rt], b, acc[rt][ct]);
                }
            }
        }
    };

    // One step: the lhs DMA, the probabilities and the x chunk of step k+1 are issued during step k and retired by the
    // wait at the top of step k+1.  (Fetching x two steps ahead -- a second chunk register set, counted vmcnt -- was
    // measured neutral: per step the vector phase, the LDS/MFMA phase and the barriers add up; nothing waits on memory.)
    auto do_step = [&](int64_t base, int cur) {
        const bool more1 = base + TRB < rend;
        const int nvalid_cur = (int)((rend - base) < TRB ? (rend - base) : TRB);
        TN_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TN_STAMP(1);
#pragma unroll
        for (int i = 0; i < PLN; ++i) {
            asm volatile("" : "+v"(plr[i]));
            const int idx = threadIdx.x + 1024 * i;
            const float pv = probs_on(base, i) ? plr[i] : 0.f;
            if (idx < TRB * MAXS * M_) pl[idx] = f32x2{pv, pv};
        }
#pragma unroll
        for (int m = 0; m < M_; ++m) asm volatile("" : "+v"(Rb[m]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TN_STAMP(2);
        __builtin_amdgcn_s_barrier();                     // MFMAs of the previous step done; lhs tile + probabilities visible
        TN_STAMP(3);
        if (more1) {
#ifndef AECF_ABL_TN_NODMA
            issue_dma(base + TRB, cur ^ 1);
#endif
            load_probs(base + TRB);
        }
        const f32x2* plc = pl + prow * (MAXS * M_);
        f32x2 xv[M_][4];
#pragma unroll
        for (int m = 0; m < M_; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xv[m][i] = f32x2{__uint_as_float(Rb[m][i] << 16), __uint_as_float(Rb[m][i] & 0xffff0000u)};
#pragma unroll
        for (int m = 0; m < M_; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xv[m][i]));
#ifndef AECF_ABL_TN_NOX
        if (more1) load_x(base + TRB);                    // the chunk registers are free again: next step's chunk flies
#endif
        const int woff = tr_off(prow, pc);
        // the slot loop in two copies picked by ONE branch per step: p.pool_lo tested per element made hipcc emit 16 branches and
        // the dead subtract / convert chains inside this phase (the phase every wave of the block waits for at the barrier below)
        auto pool_slots = [&](auto lo_tag) {
            constexpr bool LO = decltype(lo_tag)::value;
#pragma unroll
            for (int sl = 0; sl < MAXS; ++sl) {
                // M <= 3: every one of the MAXS slots, no test against nslots (a slot the tile does not have carries zero
                // probabilities): straight-line code.  M = 4 keeps the test: without it the body spills at the 128-VGPR cap
                if (M_ >= 4 && sl >= nslots) continue;
                f32x2 pv[4];
                const f32x2 p0 = plc[sl * M_];
#pragma unroll
                for (int i = 0; i < 4; ++i) pv[i] = xv[0][i] * p0;
#pragma unroll
                for (int m = 1; m < M_; ++m) {
                    const f32x2 pm = plc[sl * M_ + m];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pv[i] = __builtin_elementwise_fma(xv[m][i], pm, pv[i]);
                }
                u32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (LO) {                                      // the low parts of the pooled values (AECF_HILO_GRADS)
                        pv[i][0] -= X::to_f32(X::from_f32(pv[i][0]));
                        pv[i][1] -= X::to_f32(X::from_f32(pv[i][1]));
                    }
                    o[i] = pack_bf16x2(pv[i][0], pv[i][1]);
                }
                *reinterpret_cast<u32x4*>(ldsR + sl * TR_TILE + woff) = o;
            }
        };
#ifdef AECF_ABL_TN_NOPOOL
        if (base == rbeg)
#endif
        {
            if (p.pool_lo) pool_slots(std::true_type()); else pool_slots(std::false_type());
        }
        TN_STAMP(4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TN_STAMP(5);
        __builtin_amdgcn_s_barrier();                     // pooled tiles visible
        TN_STAMP(6);
#ifdef AECF_ABL_TN_NOMMA
        if (base == rbeg)
#endif
        mma_phase(cur, nvalid_cur);
        TN_STAMP(7);
#ifdef AECF_TN_TIMELINE
        ++step_no;
#endif
    };

    if (rbeg < rend) {
        issue_dma(rbeg, 0);
        load_probs(rbeg);
        load_x(rbeg);
    }
    int cur = 0;
    for (int64_t base = rbeg; base < rend; base += TRB, cur ^= 1) do_step(base, cur);

    float* out = p.out + (int64_t)split * EJ * E;
    if (wave_on) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (k0w + 16 * ct < kcols) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[(int64_t)(j0 + j0w + 16 * rt + 4 * lg + r) * E + k0 + k0w + 16 * ct + r16] = acc[rt][ct][r];
                }
        if (do_cs && r16 == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    p.colsum[(int64_t)split * EJ + j0 + j0w + 16 * rt + 4 * lg + r] = csacc[rt][r];
        }
    }
}

template <int M_, int MAXS>
void launch_wide(const GemmTnArgs& a, hipStream_t s) {
    size_t smem = (size_t)(4 + MAXS) * TR_TILE + (size_t)TRB * MAXS * M_ * 2 * sizeof(float);
    const int EJ = a.Ej > 0 ? a.Ej : a.E;
    dim3 grid(xcd_grid((unsigned)a.splits, (unsigned)(((a.E + 127) / 128) * ((EJ + 255) / 256)))), block(1024);
    auto kern = gemm_tn_tr_wide_kernel<M_, MAXS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

template <int M_, bool POOLED, int MAXS>
void launch_one(const GemmTnArgs& a, hipStream_t s) {
    size_t smem = (size_t)2 * TR_TILE + (size_t)(POOLED ? MAXS : 2) * TR_TILE;
    if (POOLED) smem += (size_t)TRB * MAXS * M_ * 2 * sizeof(float);
    const int nJ = ((a.Ej > 0 ? a.Ej : a.E) + 127) / 128;
    dim3 grid(xcd_grid((unsigned)a.splits, (unsigned)(((a.E + 127) / 128) * nJ))), block(512);
    auto kern = gemm_tn_tr_kernel<M_, POOLED, MAXS>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// u[h][k] = sum_{b,m} ds[b,h,m] x[b,m,k]  as a streaming kernel on the matrix pipe (round 4; bf16, H <= 8): the key-side batch
// reduction of the shapes whose score gradient does not form it (d = 768 / 1024).  u_stream_kernel did it with float32 FMAs
// on register-resident rows (8 H M FMAs per 16 bytes of x: 43 us for the 134 MB of the configs[4] shard, 3.1 TB/s); here the
// rows are the K index of an MFMA whose A operand is read TRANSPOSED out of an LDS tile (the u phase of dsu_ws_kernel as
// its own kernel): u^T[k, slot] += x^T[k, (b,m)] dsop[(b,m), slot], slots 0-7 = bf16 hi of ds per head, 8-15 = lo, added
// at the end.  Block = 4 waves, a [32 flat rows][128 columns] tile per step by LDS-DMA, three buffers (two tiles in
// flight), one barrier per step; grid = (E / 128 column tiles) x (batch splits): many small blocks per CU, so the kernel is
// bound by the row stream alone.
template <int NW>
__global__ __launch_bounds__(64 * NW) void u_mfma_kernel(GemmTnArgs p) {
    using X = Tr<BF16>;
    constexpr int NT = 64 * NW, ROWB = 64 * NW, CPR = 4 * NW;      // threads, tile row bytes (32 columns per wave), chunks per row
    constexpr int TB = 32 * ROWB;                                  // bytes of one x tile
    constexpr int DSS = 40;                                        // ds operand row: 32 K slots + 8 (rows 5 x 16 B apart)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xt = smem;                                               // [3][32][ROWB]
    unsigned short (*dsop)[16 * DSS] = reinterpret_cast<unsigned short (*)[16 * DSS]>(smem + 3 * TB);   // [2][16][DSS]
    const int E = p.E, H = p.H, M = p.M;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int k0 = blockIdx.x * (32 * NW);
    const int split = blockIdx.y;
    const int64_t fbeg = (int64_t)split * p.u_rows_per_split * M;   // flat (b, m) rows of this split
    int64_t fend = fbeg + p.u_rows_per_split * M;
    if (fend > p.B * M) fend = p.B * M;
    if (fbeg >= fend) return;
    const char* xsrc = reinterpret_cast<const char*>(p.rhs) + (int64_t)k0 * 2;
    const unsigned int ldx = (unsigned)E * 2u;

    for (int i = threadIdx.x; i < 2 * 16 * DSS / 2; i += NT) reinterpret_cast<unsigned int*>(smem + 3 * TB)[i] = 0u;   // (padding stays 0)
    // tile image: 256-byte rows, chunk ch of row r at ch ^ key(r), key = ((r & 3) << 2) | ((r >> 2) & 3)  (T10, image (b)):
    // conflict-free for the transposed reads below; the DMA destination is lane-linear, the permutation goes on the source
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    auto issue_tile = [&](int64_t f0, int buf) {
        const int nv = (int)((fend - f0) < 32 ? (fend - f0) : 32);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = threadIdx.x + NT * i;
            const int row = c / CPR, pch = c - row * CPR;
            const int key = ((row & 3) << 2) | ((row >> 2) & 3);     // (on the chunk's low 4 bits: inside its 256-byte segment)
            const int rowc = row < nv ? row : nv - 1;
            const unsigned int voff = (unsigned)rowc * ldx + (unsigned)((pch ^ key) << 4);
            const unsigned int dst = (unsigned)(size_t)(lds_void_t*)(xt + buf * TB + (64 * w + NT * i) * 16);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(xsrc + f0 * (int64_t)ldx), "s"(dst)
                         : "memory", "m0");
        }
    };
#pragma clang diagnostic pop
    // ds staging role: flat row fr = t >> 3 of the step, head t & 7
    const int fr = (threadIdx.x >> 3) & 31, hh = threadIdx.x & 7;     // (threads >= 256 repeat the first 256: same values, same slots)
    float dsv = 0.f;
    auto load_ds = [&](int64_t f0) {
        const int64_t f = f0 + fr;
        const int64_t fc = f < fend ? f : fend - 1;
        const int64_t b = fc / M;
        const int m = (int)(fc - b * M);
        const float* src = p.dsbuf + (b * H + (hh < H ? hh : 0)) * M + m;
        asm volatile("global_load_dword %0, %1, off" : "+v"(dsv) : "v"(src) : "memory");
    };
    // transposed-read addresses (constant): K slot 8 lg + 4 hi + q = tile row, columns 32 w + 16 ct + 4 pp .. + 3
    const int q = r16 >> 2, pp = r16 & 3;
    int ta[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int row = 8 * lg + 4 * hi + q;
            const int key = ((row & 3) << 2) | ((row >> 2) & 3);
            const int ch = 4 * w + 2 * ct + (pp >> 1);
            ta[ct][hi] = ROWB * row + ((ch ^ key) << 4) + 8 * (pp & 1);
        }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    issue_tile(fbeg, 0);
    load_ds(fbeg);
    if (fbeg + 32 < fend) issue_tile(fbeg + 32, 1);
    int buf = 0, step = 0;
    for (int64_t f0 = fbeg; f0 < fend; f0 += 32, buf = buf == 2 ? 0 : buf + 1, ++step) {
        // tile(step) and ds(step) landed: behind them only the 2 copies of tile(step + 1) may still fly
        if (f0 + 32 < fend) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(dsv));
        {
            const bool on = f0 + fr < fend && hh < H;
            const float d = on ? dsv : 0.f;
            const unsigned short hi = X::from_f32(d);
            if (threadIdx.x < 256) {
                dsop[step & 1][hh * DSS + fr] = hi;
                dsop[step & 1][(8 + hh) * DSS + fr] = X::from_f32(d - X::to_f32(hi));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // tile + operand of this step visible; the previous step's reads done
        if (f0 + 32 < fend) load_ds(f0 + 32);
        if (f0 + 64 < fend) issue_tile(f0 + 64, buf == 0 ? 2 : buf - 1);
        const u32x4 bop = *reinterpret_cast<const u32x4*>(&dsop[step & 1][r16 * DSS + 8 * lg]);
        const char* tile = xt + buf * TB;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const u32x4 af = tr_frag(tile, ta[ct][0], ta[ct][1]);
            acc[ct] = X::mma(af, bop, acc[ct]);
        }
    }
    // slots r16 (hi) and r16 + 8 (lo) of a head meet; lane (lg, r16 = head) stores columns k0 + 32 w + 16 ct + 4 lg .. + 3
    float* u = p.u + (int64_t)split * H * E;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        f32x4 v = acc[ct];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], 8, 64);
        if (r16 < H && r16 < 8) *reinterpret_cast<f32x4*>(u + (int64_t)r16 * E + k0 + 32 * w + 16 * ct + 4 * lg) = v;
    }
}

#ifndef AECF_UNW
#define AECF_UNW 0
#endif
bool u_mfma_supported(const GemmTnArgs& a) { return a.H <= 8 && a.E % 128 == 0 && a.M >= 1 && a.u_splits > 0; }

// block width: 32 columns per wave; the widest block whose columns tile E (each x row is then read in the fewest pieces)
void launch_u_mfma(const GemmTnArgs& a, hipStream_t s) {
    int nw = AECF_UNW;
    if (nw == 0) nw = a.E % 512 == 0 ? 16 : (a.E % 256 == 0 ? 8 : 4);
    dim3 grid((unsigned)(a.E / (32 * nw)), (unsigned)a.u_splits);
    const size_t smem = (size_t)3 * 32 * 64 * nw + (size_t)2 * 16 * 40 * 2;
    if (nw == 16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(u_mfma_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        u_mfma_kernel<16><<<grid, dim3(1024), smem, s>>>(a);
    } else if (nw == 8) {
        u_mfma_kernel<8><<<grid, dim3(512), smem, s>>>(a);
    } else {
        u_mfma_kernel<4><<<grid, dim3(256), smem, s>>>(a);
    }
}

// head slots a block needs = the most heads any aligned 128-row window of the E output rows touches
static int max_slots_128(int E, int hd) {
    int mx = 1;
    for (int j0 = 0; j0 < E; j0 += 128) {
        const int j1 = (j0 + 128 < E ? j0 + 128 : E) - 1;
        const int n = j1 / hd - j0 / hd + 1;
        if (n > mx) mx = n;
    }
    return mx;
}

// bf16 only; head_dim % 32 == 0 (a wave's 32 output rows lie inside one head)
// head slots of the widest aligned 256-row window
static int max_slots_256(int E, int hd) {
    int mx = 1;
    for (int j0 = 0; j0 < E; j0 += 256) {
        const int j1 = (j0 + 256 < E ? j0 + 256 : E) - 1;
        const int n = j1 / hd - j0 / hd + 1;
        if (n > mx) mx = n;
    }
    return mx;
}

void launch_gemm_tn_tr(const GemmTnArgs& a, hipStream_t s) {
    if (!a.pooled) { launch_one<1, false, 1>(a, s); return; }
    // 256-row tiles (1024 threads) when they tile E exactly with at most 4 head slots and M <= 3 (128-VGPR budget)
    if (!env_no_wide_tn() && a.Ej <= 0 && a.E % 256 == 0 && max_slots_256(a.E, a.hd) <= 4 &&
        (a.M <= 3 || (a.M == 4 && max_slots_256(a.E, a.hd) <= 2))) {
        const bool two = max_slots_256(a.E, a.hd) <= 2;
        switch (a.M) {
            case 1: if (two) launch_wide<1, 2>(a, s); else launch_wide<1, 4>(a, s); return;
            case 2: if (two) launch_wide<2, 2>(a, s); else launch_wide<2, 4>(a, s); return;
            case 3: if (two) launch_wide<3, 2>(a, s); else launch_wide<3, 4>(a, s); return;
            default: launch_wide<4, 2>(a, s); return;
        }
    }
    const int ns = max_slots_128(a.E, a.hd);
    AECF_DISPATCH_M(a.M, {
        if (ns <= 2) launch_one<M_, true, 2>(a, s);
        else launch_one<M_, true, 4>(a, s);
    });
}

}  // namespace aecf

#ifdef AECF_TN_TIMELINE
extern "C" int aecf_debug_tn_timeline(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(aecf::g_tn_timeline), sizeof(unsigned long long) * (size_t)n);
}
#endif
