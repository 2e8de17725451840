// Contrastive term (InfoNCE) on three tile GEMMs, bf16, gfx950 -- the fast path of aecf_nce_fwd_bwd / aecf_nce_sym_*.
//
// Local rows a [R, d] against all (gathered) keys b [C, d], unit-norm rows, positives at column off + i:
//   pass 1   E[i, j] = exp((a_i.b_j - 1) / T)   bf16 [Rp, Cp] in the workspace, stored as [Rp/256][Cp/64] tiles of [256][64]
//            (32 KB each, contiguous: the copy of an operand tile of either product reads whole DRAM pages instead of 128-
//            or 512-byte pieces of rows 128 KB apart) (the shift 1/T bounds every logit of unit-norm
//            rows, so no running maximum is needed and COLUMN sums are meaningful across row blocks and across ranks);
//            row sums l_i and column sums c_j of E come out of the same epilogue (float32, fixed-order partials).
//   pass 2   W[i, j] = coef/T (E_ij (1/l_i + sym/c_j) - (1 + sym) [j = off + i])          in place over E
//            da = W b   (reduction over the keys, split over blocks, float32 slabs reduced in fixed order)
//            db = W^T a (reduction over the local rows)
// sym = 0 is one InfoNCE direction (row softmax); sym = 1 adds the other direction of the symmetric loss from the SAME
// logits block: its softmax runs down the columns, whose sums are the only thing ranks have to exchange (one all-reduce of
// C floats between the passes).  2 R C d flops for the logits + 4 R C d for the two gradient products, against 8 R C d per
// direction of the streaming form (aecf_nce_flash.hip), which stays as the O(R d)-workspace alternative.
//
// One kernel, three operand arrangements.  Block = 512 threads = 8 waves as 2 (m) x 4 (n), block tile 256 x 256, K-step 64,
// wave tile 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16; both operand tiles arrive by LDS-DMA
// (global_load_lds_dwordx4, two stages of 2 x 32 KB), one raw s_barrier per K-step, the copy of step t + 1 flies behind the
// MFMAs of step t.  An operand whose K index is the fast axis of its source (OP_ROW: a, b in pass 1, W in da) is a
// [256 rows][128 B] tile read with ds_read_b128; one whose K index is the slow axis (OP_COL: b in da, W and a in db) is a
// [64 k][512 B] tile read TRANSPOSED with ds_read_b64_tr_b16 -- no transposed copy of W or of the embeddings exists.
// The product is formed transposed (the n operand is the MFMA A operand), so a lane ends with 4 consecutive output columns
// of one row: 8-byte bf16 / 16-byte float32 stores.
#include <math.h>
#include <type_traits>

#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

namespace {

enum { OP_ROW = 0, OP_COL = 1, OP_COLB = 2 };      // OP_COLB: OP_COL from the tiled E (m operand of db)
enum { EPI_EXP = 0, EPI_OUT = 1 };
enum { MAP_2D = 0, MAP_UNITS = 1, MAP_SPLITX = 2 };

constexpr int BT = 256;                 // block tile (m and n)
constexpr int OPB = 32768;              // bytes of one operand tile (256 x 64 bf16)
constexpr int STAGE = 2 * OPB;

struct NceGemmArgs {
    const char* a;                      // m operand
    const char* b;                      // n operand
    unsigned int lda, ldb;              // source row pitch, bytes
    int64_t a_sm, a_st;                 // OP_ROW m operand: origin of tile (mi, t) = a + mi a_sm + t a_st;  OP_COLB: a_sm = tiles per tile row
    int a_rows, b_rows;                 // source rows that exist (the rest re-read the last one)
    int a_cbytes, b_cbytes;             // OP_COL: bytes of a source row that exist (multiple of 16; the rest re-read chunk 0)
    int m_tiles, n_tiles, k_steps;      // output tiles, K / 64
    int splits, steps_per_split;        // MAP_UNITS: K range of a block
    int m_valid, n_valid;               // output rows / columns that exist
    // EPI_EXP
    unsigned short* e;                  // [m_tiles][e_tiles][256][64]
    int64_t e_tiles;                    // tiles of 64 columns per tile row = Cp / 64
    float scale2, shift2;               // E = exp2(acc * scale2 - shift2)
    float* rowsum_part;                 // [n_tiles][m_tiles 256]
    float* colsum_part;                 // [m_tiles][n_tiles 256]
    // EPI_OUT
    void* out;                          // [splits][m_valid][ldo] float32, or (out_bf16, no splits) bf16
    int64_t ldo, slab_stride;
    int out_bf16;
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const char* src, unsigned int voff, char* lds_dst) {
    const unsigned int dst = (unsigned)(size_t)(lds_void_t*)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop

// [256 rows][128 B] tile (OP_ROW), 16-byte chunk p of row r at chunk p ^ ((r >> 1) & 7): 16 consecutive rows x one logical chunk
// cover the 16 slots of a 256-byte bank row (conflict-free ds_read_b128).  The DMA destination is lane-linear, so the
// permutation goes on the source address.  A tile is 4 pieces (one wave-instruction per wave each); piece i of thread tid
// copies chunk c = tid + 512 i: row (tid >> 3) + 64 i, physical chunk tid & 7.  Rows >= rows_valid re-read the last one.
//
// [64 k][512 B] tile (OP_COL) = two [64][256 B] images of 8-row x 32-column subtiles (cdna_hip_programming.md T10, image (a)):
//   off(r, ch) = 2048 (r >> 3) + 512 (ch >> 2) + 64 (r & 7) + 16 ((ch & 3) ^ ((r >> 2) & 3))
// chunk c = tid + 512 i of image im lands at byte 16 c = row 8 (c >> 7) + ((c >> 2) & 7), chunk 4 ((c >> 5) & 3) + ((c & 3) ^ ((row >> 2) & 3));
// piece = 2 im + i.  cbytes = bytes of the source row that exist from the tile's first column on (chunks past it re-read chunk 0).
struct OperandSrc {
    const char* src;                    // origin of the tile (wave-uniform)
    int rows_valid;
};

template <int MODE, int PIECE>
__device__ __forceinline__ void issue_piece(const OperandSrc& o, unsigned int ld, int cbytes, char* lds) {
    const int tid = threadIdx.x;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    if (MODE == OP_ROW) {
        const int row = (tid >> 3) + 64 * PIECE;
        const int rowc = row < o.rows_valid ? row : o.rows_valid - 1;
        const unsigned int voff = (unsigned)rowc * ld + (unsigned)(((tid & 7) ^ ((tid >> 4) & 7)) << 4);
        dma16(o.src, voff, lds + (wbase + 512 * PIECE) * 16);
    } else {
        constexpr int im = PIECE >> 1, i = PIECE & 1;
        const int row = 8 * (tid >> 7) + 32 * i + ((tid >> 2) & 7);
        const int rowc = row < o.rows_valid ? row : o.rows_valid - 1;
        int cb = 256 * im + 16 * (4 * ((tid >> 5) & 3) + ((tid & 3) ^ ((row >> 2) & 3)));
        if (MODE == OP_COLB) {
            // columns 64 jb .. 64 jb + 63 of a tile row live in tile jb: [256 rows][128 B], 32 KB apart
            dma16(o.src, (unsigned)row * 128u + (unsigned)(cb >> 7) * 32768u + (unsigned)(cb & 127),
                  lds + 16384 * im + (wbase + 512 * i) * 16);
        } else {
            cb = cb < cbytes ? cb : 0;
            dma16(o.src, (unsigned)rowc * ld + (unsigned)cb, lds + 16384 * im + (wbase + 512 * i) * 16);
        }
    }
}

// block id (virtual: a block of the logits pass walks several) -> (m tile, n tile, K split); false = padding id
template <int MAP>
__device__ __forceinline__ bool nce_tile_of(const NceGemmArgs& p, unsigned int vb, int& mi, int& ni, int& split) {
    const unsigned int x = vb & 7u, s = vb >> 3;                // blocks b and b + 8 share an XCD (its L2)
    split = 0;
    if (MAP == MAP_2D) {
        // 32 consecutive blocks of an XCD form a 4 (m) x 8 (n) patch of tiles: 12 operand panels serve 32 tiles
        const unsigned int nsm = (p.m_tiles + 3) / 4, nsn = (p.n_tiles + 7) / 8;
        const unsigned int T = (s >> 5) * 8u + x, wi = s & 31u;
        if (T >= nsm * nsn) return false;
        mi = (int)((T / nsn) * 4 + (wi & 3));
        ni = (int)((T % nsn) * 8 + (wi >> 2));
        return mi < p.m_tiles && ni < p.n_tiles;
    } else if (MAP == MAP_SPLITX) {
        // 8 K splits, one per XCD: every block of an XCD walks the same K range, so the n operand's K slices are shared by
        // all of them through its L2; the n tiles of an m tile are neighbours (the m operand is fetched once)
        ni = (int)(s % p.n_tiles);
        mi = (int)(s / p.n_tiles);
        split = (int)x;
        return mi < p.m_tiles;
    } else {
        // the n tiles of one (m tile, K split) are neighbours on one XCD: the big operand is fetched from HBM once
        const unsigned int unit = (s / p.n_tiles) * 8u + x;
        if (unit >= (unsigned)(p.m_tiles * p.splits)) return false;
        ni = (int)(s % p.n_tiles);
        mi = (int)(unit % p.m_tiles);
        split = (int)(unit / p.m_tiles);
        return true;
    }
}

template <int AM, int BM, int EPI, int MAP>
__global__ __launch_bounds__(512, 2) void nce_gemm_kernel(NceGemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int wm = w >> 2, wn = w & 3;

    // ---- fragment addresses
    int a_row[2], b_row[2];             // OP_ROW: per K-step of 32
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ((4 * ks + lg) ^ (r16 >> 1)) & 7;
        a_row[ks] = (128 * wm + r16) * 128 + (ch << 4);
        b_row[ks] = (64 * wn + r16) * 128 + (ch << 4);
    }
    // OP_COL: lane 4 q + pp of group lg reads row 32 ks + 8 lg + 4 hh + q, columns 4 pp .. 4 pp + 3 of 16-column block blk:
    //   8192 ks + 2048 lg + 512 (blk >> 1) + 256 hh + 64 q + 16 ((2 (blk & 1) + (pp >> 1)) ^ (2 (lg & 1) + hh)) + 8 (pp & 1)
    const int q = r16 >> 2, pp = r16 & 3;
    int tx[2][2];
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            tx[b1][hh] = 2048 * lg + 64 * q + 8 * (pp & 1) + 256 * hh + 16 * ((2 * b1 + (pp >> 1)) ^ (2 * (lg & 1) + hh));
    const int a_img = 16384 * wm;                               // m blocks 8 wm + rt: image wm, block rt
    const int b_img = 16384 * (wn >> 1) + 512 * (2 * (wn & 1)); // n blocks 4 wn + ct: image wn >> 1, block 4 (wn & 1) + ct

    // operand tile of K-step t of output tile (mi, ni)
    auto src_a = [&](int t, int mi) -> OperandSrc {
        if (AM == OP_ROW) return OperandSrc{p.a + mi * p.a_sm + t * p.a_st, p.a_rows - BT * mi};
        if (AM == OP_COLB)      // K rows 64 t .. of tile row t / 4, columns of tiles 4 mi .. 4 mi + 3
            return OperandSrc{p.a + ((int64_t)(t >> 2) * p.a_sm + 4 * (int64_t)mi) * 32768 + (t & 3) * 8192, 64};
        const int k0 = 64 * t < p.a_rows ? 64 * t : p.a_rows - 1;
        return OperandSrc{p.a + (int64_t)k0 * p.lda + 512 * (int64_t)mi, 64 * t < p.a_rows ? p.a_rows - 64 * t : 1};
    };
    auto src_b = [&](int t, int ni) -> OperandSrc {
        if (BM == OP_ROW) return OperandSrc{p.b + (int64_t)BT * ni * p.ldb + 128 * (int64_t)t, p.b_rows - BT * ni};
        const int k0 = 64 * t < p.b_rows ? 64 * t : p.b_rows - 1;
        return OperandSrc{p.b + (int64_t)k0 * p.ldb + 512 * (int64_t)ni, 64 * t < p.b_rows ? p.b_rows - 64 * t : 1};
    };
#define NCE_PIECE(P_, oa_, ob_, stage_, mi_, ni_)                                                                       \
    do {                                                                                                                \
        if ((P_) < 4) issue_piece<AM, (P_) & 3>(oa_, p.lda, p.a_cbytes - 512 * (mi_), smem + (stage_) * STAGE);          \
        else issue_piece<BM, (P_) & 3>(ob_, p.ldb, p.b_cbytes - 512 * (ni_), smem + (stage_) * STAGE + OPB);             \
    } while (0)
    // fragment of slot sl = 8 ks + rt (m operand) / of (ks, ct) (n operand) from the tile at lds
    auto read_a = [&](const char* la, int sl) -> u32x4 {
        const int ks = sl >> 3, rt = sl & 7;
        if (AM == OP_ROW) return *reinterpret_cast<const u32x4*>(la + a_row[ks] + 2048 * rt);
        const int o = a_img + 8192 * ks + 512 * (rt >> 1);
        return tr_frag16(la, o + tx[rt & 1][0], o + tx[rt & 1][1]);
    };
    auto read_b = [&](const char* lb, int ks, int ct) -> u32x4 {
        if (BM == OP_ROW) return *reinterpret_cast<const u32x4*>(lb + b_row[ks] + 2048 * ct);
        const int o = b_img + 8192 * ks + 512 * (ct >> 1);
        return tr_frag16(lb, o + tx[ct & 1][0], o + tx[ct & 1][1]);
    };
    auto k_range = [&](int split, int& t_beg, int& t_end) {
        t_beg = split * p.steps_per_split;
        t_end = t_beg + p.steps_per_split < p.k_steps ? t_beg + p.steps_per_split : p.k_steps;
    };
    OperandSrc pa = {nullptr, 1}, pb = {nullptr, 1};            // the copy whose pieces 3..7 are still to be issued
    // first copies of an output tile: step t_beg whole into stage 0, the first 3 pieces of step t_beg + 1 into stage 1
    auto issue_first = [&](int mi, int ni, int t_beg, int t_end) {
        if (t_beg >= t_end) return;
        const OperandSrc oa = src_a(t_beg, mi), ob = src_b(t_beg, ni);
        NCE_PIECE(0, oa, ob, 0, mi, ni); NCE_PIECE(1, oa, ob, 0, mi, ni); NCE_PIECE(2, oa, ob, 0, mi, ni);
        NCE_PIECE(3, oa, ob, 0, mi, ni); NCE_PIECE(4, oa, ob, 0, mi, ni); NCE_PIECE(5, oa, ob, 0, mi, ni);
        NCE_PIECE(6, oa, ob, 0, mi, ni); NCE_PIECE(7, oa, ob, 0, mi, ni);
        if (t_beg + 1 < t_end) {
            pa = src_a(t_beg + 1, mi); pb = src_b(t_beg + 1, ni);
            NCE_PIECE(0, pa, pb, 1, mi, ni); NCE_PIECE(1, pa, pb, 1, mi, ni); NCE_PIECE(2, pa, pb, 1, mi, ni);
        }
    };

    int mi = 0, ni = 0, split = 0, t_beg = 0, t_end = 0;
    if (!nce_tile_of<MAP>(p, blockIdx.x, mi, ni, split)) return;
    k_range(split, t_beg, t_end);
    issue_first(mi, ni, t_beg, t_end);
    {
        f32x4 acc[8][4];
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- main loop.  A K-step is 16 slots of 4 MFMAs (slot = (ks, rt): one m fragment against the 4 n fragments); the m
        // fragments run through a ring of 4 registers, read 3 slots ahead.  ONE barrier per K-step, at slot 13: by then every
        // read of this step's stage has been issued (and is waited for), and the copy of step t + 1 -- issued a full step
        // earlier -- is waited for, so behind the barrier (a) slots 13..15 read the first fragments of step t + 1 from the
        // other stage (no bubble at the step boundary) and (b) the copy of step t + 2 into THIS stage starts.  The 8
        // wave-instructions of a copy are spread over 8 slots (3 behind the barrier, 5 at the start of the next step) so that
        // their issue cost hides behind MFMAs instead of stacking up in front of them.
        u32x4 af[4], bf0[4], bf1[4];
        if (t_beg < t_end) {
            // younger than the first step's 8 pieces: the 3 pieces of the second step
            if (t_beg + 1 < t_end) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            af[0] = read_a(smem, 0); af[1] = read_a(smem, 1); af[2] = read_a(smem, 2);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) bf0[ct] = read_b(smem + OPB, 0, ct);
        }
        int st = 0;
        // one K-step; H1 / H2: steps t + 1 / t + 2 exist (compile-time: the steady-state body has no branches)
        auto kstep = [&](int t, auto h1, auto h2) {
            constexpr bool H1 = decltype(h1)::value, H2 = decltype(h2)::value;
            const char* cur = smem + st * STAGE;
            const char* nxt = smem + (st ^ 1) * STAGE;
            OperandSrc qa = {nullptr, 1}, qb = {nullptr, 1};
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) {
                if (sl == 13 && H1) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (H2) { qa = src_a(t + 2, mi); qb = src_b(t + 2, ni); }
                }
                if (sl < 5 && H1) {
                    switch (sl) {
                        case 0: NCE_PIECE(3, pa, pb, st ^ 1, mi, ni); break;
                        case 1: NCE_PIECE(4, pa, pb, st ^ 1, mi, ni); break;
                        case 2: NCE_PIECE(5, pa, pb, st ^ 1, mi, ni); break;
                        case 3: NCE_PIECE(6, pa, pb, st ^ 1, mi, ni); break;
                        default: NCE_PIECE(7, pa, pb, st ^ 1, mi, ni); break;
                    }
                }
                if (sl >= 13 && H2) {
                    switch (sl) {
                        case 13: NCE_PIECE(0, qa, qb, st, mi, ni); break;
                        case 14: NCE_PIECE(1, qa, qb, st, mi, ni); break;
                        default: NCE_PIECE(2, qa, qb, st, mi, ni); break;
                    }
                }
                if (sl <= 12) af[(sl + 3) & 3] = read_a(cur, sl + 3);
                else if (H1) af[(sl + 3) & 3] = read_a(nxt, sl - 13);
                if (sl >= 4 && sl <= 7) bf1[sl - 4] = read_b(cur + OPB, 1, sl - 4);
                if (sl >= 13 && H1) {
                    bf0[sl - 13] = read_b(nxt + OPB, 0, sl - 13);
                    if (sl == 15) bf0[3] = read_b(nxt + OPB, 0, 3);
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[sl & 7][ct] = Tr<BF16>::mma(sl < 8 ? bf0[ct] : bf1[ct], af[sl & 3], acc[sl & 7][ct]);
                __builtin_amdgcn_sched_barrier(0);              // the slot order IS the schedule
            }
            pa = qa; pb = qb;
            st ^= 1;
        };
        {
            using T_ = std::integral_constant<bool, true>;
            using F_ = std::integral_constant<bool, false>;
            int t = t_beg;
            for (; t + 2 < t_end; ++t) kstep(t, T_{}, T_{});
            if (t + 1 < t_end) { kstep(t, T_{}, F_{}); ++t; }
            if (t < t_end) kstep(t, F_{}, F_{});
        }

        // ---- epilogue: lane (r16, lg) holds C[m = 128 wm + 16 rt + r16][n = 64 wn + 16 ct + 4 lg + r], r = 0..3
        const int64_t gi0 = (int64_t)BT * mi + 128 * wm + r16;
        const int gj0 = BT * ni + 64 * wn + 4 * lg;
        if (EPI == EPI_OUT) {
            float* o = reinterpret_cast<float*>(p.out) + (int64_t)split * p.slab_stride;
            unsigned short* ob = reinterpret_cast<unsigned short*>(p.out);
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                const int64_t i = gi0 + 16 * rt;
                if (i < p.m_valid) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const int j = gj0 + 16 * ct;
                        if (j < p.n_valid) {
                            if (p.out_bf16)
                                *reinterpret_cast<u32x2*>(ob + i * p.ldo + j) =
                                    u32x2{pack_bf16x2(acc[rt][ct][0], acc[rt][ct][1]), pack_bf16x2(acc[rt][ct][2], acc[rt][ct][3])};
                            else *reinterpret_cast<f32x4*>(o + i * p.ldo + j) = acc[rt][ct];
                        }
                    }
                }
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // every wave is past its last read of the stages
            // E = exp2(acc scale - shift): 4 consecutive columns per lane, 8-byte stores (16 rows x 32 B per wave-instruction).
            // The sums run as packed float32 adds (both halves from their own registers); masking only on edge tiles.
            float rs[8];
            f32x2 cs2[4][2];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) cs2[ct][0] = cs2[ct][1] = f32x2{0.f, 0.f};
            float* lrs = reinterpret_cast<float*>(smem);        // [4][256] row sums | [2][256] column sums
            float* lcs = lrs + 4 * BT;
            const bool edge = BT * (mi + 1) > p.m_valid || BT * (ni + 1) > p.n_valid;      // block-uniform
            unsigned short* etile = p.e + ((int64_t)mi * p.e_tiles + 4 * ni + wn) * (BT * 64) + (128 * wm + r16) * 64 + 4 * lg;
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                const int64_t i = gi0 + 16 * rt;
                f32x2 s2 = f32x2{0.f, 0.f};
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int j = gj0 + 16 * ct;
                    f32x2 e01, e23;
                    e01[0] = __builtin_amdgcn_exp2f(acc[rt][ct][0] * p.scale2 - p.shift2);
                    e01[1] = __builtin_amdgcn_exp2f(acc[rt][ct][1] * p.scale2 - p.shift2);
                    e23[0] = __builtin_amdgcn_exp2f(acc[rt][ct][2] * p.scale2 - p.shift2);
                    e23[1] = __builtin_amdgcn_exp2f(acc[rt][ct][3] * p.scale2 - p.shift2);
                    if (edge) {
                        const bool iok = i < p.m_valid;
                        e01[0] = (iok && j + 0 < p.n_valid) ? e01[0] : 0.f;
                        e01[1] = (iok && j + 1 < p.n_valid) ? e01[1] : 0.f;
                        e23[0] = (iok && j + 2 < p.n_valid) ? e23[0] : 0.f;
                        e23[1] = (iok && j + 3 < p.n_valid) ? e23[1] : 0.f;
                    }
                    s2 += e01 + e23;
                    cs2[ct][0] += e01;
                    cs2[ct][1] += e23;
                    // tile (mi, 4 ni + wn) of E: row 128 wm + 16 rt + r16, columns 16 ct + 4 lg .. + 3 of its 64 -- the four
                    // stores of a row group fill whole 128-byte lines of one 32 KB tile
                    *reinterpret_cast<u32x2*>(etile + (16 * rt) * 64 + 16 * ct) = u32x2{pack_bf16x2(e01[0], e01[1]), pack_bf16x2(e23[0], e23[1])};
                }
                rs[rt] = reduce_lg(s2[0] + s2[1]);              // over the wave's 64 columns
            }
            float cs[4][4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { cs[ct][0] = cs2[ct][0][0]; cs[ct][1] = cs2[ct][0][1]; cs[ct][2] = cs2[ct][1][0]; cs[ct][3] = cs2[ct][1][1]; }
            if (lg == 0) {
#pragma unroll
                for (int rt = 0; rt < 8; ++rt) lrs[wn * BT + 128 * wm + 16 * rt + r16] = rs[rt];
            }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = reduce_r16(cs[ct][r]);      // over the wave's 128 rows
                    if (r16 == 0) lcs[wm * BT + 64 * wn + 16 * ct + 4 * lg + r] = v;
                }
            __syncthreads();
            const int tdx = threadIdx.x;
            if (tdx < BT) {
                p.rowsum_part[((int64_t)ni * p.m_tiles + mi) * BT + tdx] = (lrs[tdx] + lrs[BT + tdx]) + (lrs[2 * BT + tdx] + lrs[3 * BT + tdx]);
            } else {
                const int c = tdx - BT;
                p.colsum_part[((int64_t)mi * p.n_tiles + ni) * BT + c] = lcs[c] + lcs[BT + c];
            }
        }
    }
#undef NCE_PIECE
}

// ---- small kernels around the GEMMs ------------------------------------------------------------------------------------

// l[i] = sum over the column tiles' partials, c[j] = sum over the row tiles' (fixed order: four strided partial sums per
// element, added in order).  Block = 64 elements x 4 parts.
__global__ __launch_bounds__(256) void nce_sums_kernel(const float* rowsum_part, const float* colsum_part, int m_tiles, int n_tiles,
                                                       int64_t rows, int64_t cols, float* l, float* c) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int64_t Rp = (int64_t)m_tiles * BT;
    int64_t id = (int64_t)blockIdx.x * 64 + e;                  // Rp is a multiple of 64: a block is all rows or all columns
    const bool is_row = id < Rp;
    if (!is_row) id -= Rp;
    const float* src = is_row ? rowsum_part : colsum_part;
    const int nt = is_row ? n_tiles : m_tiles, other = is_row ? m_tiles : n_tiles;
    float s = 0.f;
    for (int t = part; t < nt; t += 4) s += src[((int64_t)t * other + id / BT) * BT + id % BT];
    red[part][e] = s;
    __syncthreads();
    if (part == 0) {
        const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (is_row) { if (id < rows) l[id] = v; }
        else if (id < cols && c) c[id] = v;
    }
}

struct NceFinArgs {
    const unsigned short* a;            // [rows, d]
    const unsigned short* b;            // [cols, d]
    const float* l;                     // [rows]   row sums of E
    const float* c;                     // [cols]   column sums of E, all ranks (sym)
    float* u;                           // [Rp]     1 / l (0 in the padding)
    float* v;                           // [Cp]     1 / c (0 in the padding / sym == 0)
    float* loss_rows;                   // [rows]
    float* ediag;                       // [rows]   exp((a_i.b_pos - 1)/T) in float32: the positive's exponential before rounding
    int64_t rows, cols, row_offset, Rp, Cp;
    int d, sym;
    float inv_temp;
    const float* ent;                   // entropy regulariser riding in this launch (n_ent == 0: off)
    float* d_ent;
    float* ent_loss;
    int64_t n_ent;
    float ent_target, ent_scale;
};

// one wave per local row: u_i, the positive logit a_i.b_pos, loss_i = log l_i + 1/T - s_ii/T (+ log c_pos + 1/T - s_ii/T);
// the waves past the rows fill v; block 0 also carries CurriculumMasking.entropy_loss (ref aecf/AECFLayer.py:285-314)
__global__ __launch_bounds__(256) void nce_finalize_kernel(NceFinArgs p) {
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * 4 + wave_id();
    if (i < p.rows) {
        const unsigned short* ap = p.a + i * p.d;
        const unsigned short* bp = p.b + (p.row_offset + i) * p.d;
        float dot = 0.f;
        for (int k = lane; k < p.d; k += 64) dot = fmaf(Tr<BF16>::to_f32(ap[k]), Tr<BF16>::to_f32(bp[k]), dot);
        dot = reduce_wave(dot);
        if (lane == 0) {
            const float li = p.l[i];
            p.u[i] = 1.0f / li;
            p.ediag[i] = __builtin_amdgcn_exp2f((dot - 1.0f) * p.inv_temp * 1.4426950408889634f);
            float loss = logf(li) + p.inv_temp - dot * p.inv_temp;
            if (p.sym) loss += logf(p.c[p.row_offset + i]) + p.inv_temp - dot * p.inv_temp;
            p.loss_rows[i] = loss;
        }
    } else if (i < p.Rp) {
        if (lane == 0) p.u[i] = 0.f;
    }
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < p.Cp; j += (int64_t)gridDim.x * 256)
        p.v[j] = (p.sym && j < p.cols) ? 1.0f / p.c[j] : 0.f;
    if (blockIdx.x == 0 && p.n_ent > 0) {
        __shared__ float red[4];
        float acc = 0.f;
        for (int64_t j = threadIdx.x; j < p.n_ent; j += 256) {
            const float raw = p.ent[j];
            const float h = isnan(raw) ? 0.f : (isinf(raw) ? (raw > 0.f ? 1.f : 0.f) : raw);      // nan_to_num (ref :295-296)
            const float dlt = h - p.ent_target;
            acc += dlt * dlt;
            if (p.d_ent) p.d_ent[j] = isfinite(raw) ? p.ent_scale * dlt : 0.f;
        }
        acc = reduce_wave(acc);
        if (lane == 0) red[wave_id()] = acc;
        __syncthreads();
        if (threadIdx.x == 0) p.ent_loss[0] = fmaxf((red[0] + red[1] + red[2] + red[3]) / (float)p.n_ent, 0.f);
    }
}

// W = ct (E (u_i + v_j) - npos [j = off + i]) in place over the tiled E, 8 elements per thread.  The positive's weight is a
// small difference of O(1) terms (softmax weight minus one): it is formed from the float32 exponential, not from the bf16 one.
__global__ __launch_bounds__(256) void nce_weights_kernel(unsigned short* e, int64_t e_tiles, int64_t m_tiles, const float* u,
                                                          const float* v, const float* ediag, int64_t rows, int64_t row_offset,
                                                          float ct, float npos, const float* upstream) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;         // 16-byte chunk: 2048 per tile, 8 per tile row
    if (id >= m_tiles * e_tiles * 2048) return;
    const int64_t tile = id >> 11;
    const int within = (int)(id & 2047);
    const int64_t i = (tile / e_tiles) * BT + (within >> 3), j0 = (tile % e_tiles) * 64 + 8 * (within & 7);
    u32x4* ptr = reinterpret_cast<u32x4*>(e) + id;
    const u32x4 raw = *ptr;
    const float ui = u[i];
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(v + j0), v1 = *reinterpret_cast<const f32x4*>(v + j0 + 4);
    float x[8];
    Tr<BF16>::unpack(raw, x);
    const float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    const int64_t jp = (i < rows) ? row_offset + i - j0 : -1;
    if (upstream) ct *= upstream[0];                           // d loss / d (this call's term): a device scalar, no host read
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float wv = x[k] * (ui + vv[k]);
        if (jp == k) wv = ediag[i] * (ui + vv[k]) - npos;
        x[k] = ct * wv;
    }
    *ptr = Tr<BF16>::pack(x);
}

// out[i] = sum_s slab[s][i], float4 (rounded once to bf16 when the caller wants the gradient in that dtype)
__global__ __launch_bounds__(256) void nce_slab_sum_kernel(const float* slabs, int splits, int64_t n4, int64_t stride, void* out,
                                                           int out_bf16) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= n4) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(slabs + 4 * id);
    for (int k = 1; k < splits; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * stride + 4 * id);
        s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    if (out_bf16) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(out) + 4 * id) = u32x2{pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3])};
    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + 4 * id) = s;
}

template <int AM, int BM, int EPI, int MAP>
void launch_gemm(const NceGemmArgs& a, unsigned int blocks, hipStream_t s) {
    auto kern = nce_gemm_kernel<AM, BM, EPI, MAP>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    kern<<<dim3(blocks), dim3(512), 2 * STAGE, s>>>(a);
}

inline int64_t up256(int64_t v) { return (v + 255) / 256 * 256; }
inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

// workspace carve
struct NceWs {
    unsigned short* e;                  // [Rp][Cp] bf16
    float* rowsum_part;                 // [n_tiles][Rp]
    float* colsum_part;                 // [m_tiles][Cp]
    float* l;                           // [Rp]
    float* u;                           // [Rp]
    float* v;                           // [Cp]
    float* ediag;                       // [Rp]
    float* c_local;                     // [Cp]  column sums when the caller passes none
    float* slabs;                       // [splits][rows][d]
    size_t bytes;
};

int da_splits(int64_t Rp, int64_t Cp, int d) {
    const int64_t items = (Rp / BT) * ((d + BT - 1) / BT);
    int64_t s = (768 + items - 1) / items;              // about three blocks per CU ...
    const int64_t cap = Cp / 64 / 8;                    // ... of at least 8 K-steps each
    s = s > cap ? cap : s;
    if (s >= 5 && s <= 12 && cap >= 8) s = 8;          // one split per XCD (MAP_SPLITX)
    return (int)(s < 1 ? 1 : (s > 32 ? 32 : s));
}

NceWs carve(void* ws, int64_t rows, int64_t cols, int d) {
    const int64_t Rp = up256(rows), Cp = up256(cols);
    const int64_t mt = Rp / BT, nt = Cp / BT;
    NceWs w;
    char* p = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t n) { char* r = p + off; off += al256(n); return r; };
    w.e = (unsigned short*)take((size_t)Rp * Cp * 2);
    w.rowsum_part = (float*)take((size_t)nt * Rp * 4);
    w.colsum_part = (float*)take((size_t)mt * Cp * 4);
    w.l = (float*)take((size_t)Rp * 4);
    w.u = (float*)take((size_t)Rp * 4);
    w.v = (float*)take((size_t)Cp * 4);
    w.ediag = (float*)take((size_t)Rp * 4);
    w.c_local = (float*)take((size_t)Cp * 4);
    w.slabs = (float*)take((size_t)da_splits(Rp, Cp, d) * rows * d * 4);
    w.bytes = off;
    return w;
}

}  // namespace

bool nce_gemm_supported(int dtype, int d, float temperature) {
    // 1/T is the shift of every exponent: e^(-2/T) has to stay a normal float32 / bf16
    return dtype == 0 && d % 64 == 0 && d >= 64 && d <= 4096 && temperature >= 0.025f;
}

size_t nce_gemm_workspace_bytes(int64_t rows, int64_t cols, int d) { return carve(nullptr, rows, cols, d).bytes + 256; }

// pass 1: E, row sums (workspace) and this rank's column sums (col_sums, may be NULL when sym == 0)
void launch_nce_gemm_pass1(int64_t rows, int64_t cols, int d, float inv_temp, const void* a, const void* b, void* workspace,
                           float* col_sums, hipStream_t s) {
    const NceWs w = carve(workspace, rows, cols, d);
    const int64_t Rp = up256(rows), Cp = up256(cols);
    NceGemmArgs g = {};
    g.a = (const char*)a; g.b = (const char*)b; g.lda = g.ldb = 2u * (unsigned)d;
    g.a_sm = (int64_t)BT * g.lda; g.a_st = 128;
    g.a_rows = (int)rows; g.b_rows = (int)cols;
    g.m_tiles = (int)(Rp / BT); g.n_tiles = (int)(Cp / BT); g.k_steps = d / 64;
    g.splits = 1; g.steps_per_split = g.k_steps;
    g.m_valid = (int)rows; g.n_valid = (int)cols;
    g.e = w.e; g.e_tiles = Cp / 64;
    g.scale2 = inv_temp * 1.4426950408889634f; g.shift2 = g.scale2;
    g.rowsum_part = w.rowsum_part; g.colsum_part = w.colsum_part;
    const unsigned int nsm = (g.m_tiles + 3) / 4, nsn = (g.n_tiles + 7) / 8;
    const unsigned int blocks = ((nsm * nsn + 7) / 8) * 8 * 32;
    launch_gemm<OP_ROW, OP_ROW, EPI_EXP, MAP_2D>(g, blocks, s);
    nce_sums_kernel<<<dim3((unsigned)((Rp + Cp) / 64)), dim3(256), 0, s>>>(w.rowsum_part, w.colsum_part, g.m_tiles, g.n_tiles,
                                                                                   rows, cols, w.l, col_sums ? col_sums : w.c_local);
}

// normalisers + loss rows (col_sums: all ranks' sums when sym; NULL = pass 1's own) [+ the entropy regulariser riding along]
void launch_nce_gemm_loss(int64_t rows, int64_t cols, int64_t row_offset, int d, float inv_temp, int sym, const void* a,
                          const void* b, const float* col_sums, void* workspace, float* loss_rows, const float* ent, int64_t n_ent,
                          float ent_target, float ent_upstream, float* d_ent, float* ent_loss, hipStream_t s) {
    const NceWs w = carve(workspace, rows, cols, d);
    const int64_t Rp = up256(rows), Cp = up256(cols);
    NceFinArgs f = {};
    f.a = (const unsigned short*)a; f.b = (const unsigned short*)b; f.l = w.l; f.c = col_sums ? col_sums : w.c_local;
    f.u = w.u; f.v = w.v; f.ediag = w.ediag; f.loss_rows = loss_rows; f.rows = rows; f.cols = cols; f.row_offset = row_offset; f.Rp = Rp; f.Cp = Cp;
    f.d = d; f.sym = sym; f.inv_temp = inv_temp;
    f.ent = ent; f.d_ent = d_ent; f.ent_loss = ent_loss; f.n_ent = ent ? n_ent : 0; f.ent_target = ent_target;
    f.ent_scale = n_ent > 0 ? 2.0f * ent_upstream / (float)n_ent : 0.f;
    nce_finalize_kernel<<<dim3((unsigned)((Rp + 3) / 4)), dim3(256), 0, s>>>(f);
}

// gradients: weights in place over E (scaled by the device scalar `upstream` when given), da = W b, db = W^T a; outputs float32
// or -- one rounding of the float32 sums -- bf16.  launch_nce_gemm_loss must have run on this workspace.
void launch_nce_gemm_grads(int64_t rows, int64_t cols, int64_t row_offset, int d, float inv_temp, float coef, int sym, const void* a,
                           const void* b, void* workspace, const float* upstream, int out_bf16, void* da, void* db, hipStream_t s) {
    const NceWs w = carve(workspace, rows, cols, d);
    const int64_t Rp = up256(rows), Cp = up256(cols);
    const int64_t chunks = Rp * (Cp / 8);
    nce_weights_kernel<<<dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s>>>(w.e, Cp / 64, Rp / BT, w.u, w.v, w.ediag, rows, row_offset,
                                                                                    coef * inv_temp, sym ? 2.0f : 1.0f, upstream);
    const int n_tiles_d = (d + BT - 1) / BT;
    {   // da = W b: m = local rows, n = d, K = keys
        NceGemmArgs g = {};
        g.a = (const char*)w.e; g.lda = 128; g.a_rows = (int)Rp;          // tile (mi, t) of E: [256][128 B], contiguous
        g.a_sm = (Cp / 64) * (int64_t)32768; g.a_st = 32768;
        g.b = (const char*)b; g.ldb = 2u * (unsigned)d; g.b_rows = (int)cols; g.b_cbytes = 2 * d;
        g.m_tiles = (int)(Rp / BT); g.n_tiles = n_tiles_d; g.k_steps = (int)(Cp / 64);
        g.splits = da_splits(Rp, Cp, d);
        g.steps_per_split = (g.k_steps + g.splits - 1) / g.splits;
        g.m_valid = (int)rows; g.n_valid = d;
        g.out = g.splits > 1 ? (void*)w.slabs : da; g.ldo = d; g.slab_stride = rows * (int64_t)d;
        g.out_bf16 = g.splits > 1 ? 0 : out_bf16;
        const unsigned int units = (unsigned)(g.m_tiles * g.splits);
        if (g.splits == 8) launch_gemm<OP_ROW, OP_COL, EPI_OUT, MAP_SPLITX>(g, 8u * g.m_tiles * g.n_tiles, s);
        else launch_gemm<OP_ROW, OP_COL, EPI_OUT, MAP_UNITS>(g, ((units + 7) / 8) * 8 * g.n_tiles, s);
        if (g.splits > 1) {
            const int64_t n4 = rows * (int64_t)d / 4;
            nce_slab_sum_kernel<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s>>>(w.slabs, g.splits, n4, g.slab_stride, da, out_bf16);
        }
    }
    {   // db = W^T a: m = keys, n = d, K = local rows
        NceGemmArgs g = {};
        g.a = (const char*)w.e; g.lda = 128; g.a_rows = (int)Rp; g.a_cbytes = 0; g.a_sm = Cp / 64;
        g.b = (const char*)a; g.ldb = 2u * (unsigned)d; g.b_rows = (int)rows; g.b_cbytes = 2 * d;
        g.m_tiles = (int)(Cp / BT); g.n_tiles = n_tiles_d; g.k_steps = (int)(Rp / 64);
        g.splits = 1; g.steps_per_split = g.k_steps;
        g.m_valid = (int)cols; g.n_valid = d;
        g.out = db; g.ldo = d; g.slab_stride = 0; g.out_bf16 = out_bf16;
        const unsigned int units = (unsigned)g.m_tiles;
        launch_gemm<OP_COLB, OP_COL, EPI_OUT, MAP_UNITS>(g, ((units + 7) / 8) * 8 * g.n_tiles, s);
    }
}

}  // namespace aecf
