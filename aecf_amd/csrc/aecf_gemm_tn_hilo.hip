// The two weight-gradient batch reductions on bf16 hi + lo operand pairs, ONE launch each (AECF_HILO_GRADS, round 5):
//
//   dW_v[j][k] = sum_b do[b][j] * P(b,k)     do = do_hi + do_lo (written by the dout kernel), P = sum_m probs x  split where it is formed
//              = sum_b do_hi P_hi + do_hi P_lo + do_lo P_hi                    (the lo lo term is 2^-18 of the sum: dropped)
//   dW_o[j][k] = sum_b dy[b][j] * (o_hi[b][k] + o_lo[b][k])                    (dy is an exact bf16 input: two products)
//
// Round 4 ran each product as its own launch of the default kernels (three for dW_v, two for dW_o: 102 -> 310 us and 47 -> 107 us
// at the headline shape): every launch re-read its lhs tiles, re-fetched x, re-pooled the same rows and wrote its own slab set.
// Here a block lands the hi AND lo tiles of a step, pools once in float32, writes the pooled tile as hi + lo and issues the
// three (two) MFMAs per output fragment from the same LDS images into ONE accumulator set; the finalize launch reduces one slab
// set, as in the default path.
//
// Same tile images, transposed LDS reads (ds_read_b64_tr_b16), LDS-DMA and one-step-ahead operand fetches as
// aecf_gemm_tn_tr.hip.  What differs is the step: 32 batch rows (one MFMA K-step) instead of 64, because a step now holds
// twice the tiles -- pooled product: lhs hi + lo [32][128] x 2 buffers = 32 KB, pooled hi + lo x 2 head slots = 32 KB.
// Measured at the headline shape (profiles/r05_c2_hilo_time.txt): dW_v 92 -> 152 us, dW_o 46 -> 77 us, against 310 / 107 us
// for the round-4 launches; float32-stored parameter gradients 3 - 5e-6 of fp32 math.
#include <stdlib.h>

#include "aecf_kernels.h"
#include "aecf_tile.h"
#include "aecf_tr_tile.h"

namespace aecf {

namespace {

constexpr int HRB = 32;                        // batch rows per step
constexpr int HR_TILE = HRB * 256;             // bytes of one [32][128 bf16] tile image (the first four row groups of aecf_tr_tile.h)

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is named as a clobber on purpose
// LDS-DMA of [32 rows][256 B] images, one 16-byte chunk per thread: NT threads cover NT / 512 images that lie 256 bytes apart in
// the source row (128-column sub-tiles) and HR_TILE bytes apart in LDS.  Destination lane-linear (chunk c of an image at byte
// 16 c), the image's permutation on the per-lane source address (aecf_gemm_tn_tr.hip: dma_tile_tr_async).
__device__ __forceinline__ void dma_hr_images(const char* __restrict__ src, unsigned int ld_bytes, int rows_valid, char* lds) {
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    const int sub = wbase >> 9;                                   // (wave-uniform)
    const int c = threadIdx.x & 511;
    const int row = 8 * (c >> 7) + ((c >> 2) & 7);
    const int rowc = row < rows_valid ? row : rows_valid - 1;
    const int logical = 4 * ((c >> 5) & 3) + ((c & 3) ^ ((row >> 2) & 3));
    const unsigned int voff = (unsigned)rowc * ld_bytes + (unsigned)logical * 16u;
    const unsigned int dst = (unsigned)(size_t)(lds_void_t*)(lds + wbase * 16);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(src + 256 * sub), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop

// bf16 hi + lo of a float32 pair: hi = RNE pack, lo = RNE pack of the remainders
__device__ __forceinline__ void split_pack(float a, float b, unsigned int& hi, unsigned int& lo) {
    hi = pack_bf16x2(a, b);
    lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// ------------------------------------------------------------------------------------------------------------------
// dW_v: 512 threads = 8 waves as 4 (j) x 2 (k), block tile 128 x 128, wave tile 32 x 64, TWO blocks per CU (64 KB of LDS each
// at two head slots): the pooling phase of one block runs under the MFMA phase of the other.  (The 1024-thread 256 x 128 form
// the default path prefers was built too: 165 us against 152 us at the headline shape -- with three MFMAs per fragment pair the
// phases are long enough that overlapping them across blocks beats pooling each x slice once for four head slots.)  A thread
// pools one 16-byte chunk of a row for every head slot of the tile.
template <int M_, int MAXS>
__global__ __launch_bounds__(512, 4) void gemm_tn_hilo_pooled_kernel(GemmTnArgs p) {
    using X = Tr<BF16>;
    constexpr int RT = 2, CT = 4;
    constexpr int NPL = HRB * MAXS * M_;
    static_assert(NPL <= 512, "step shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int E = p.E, H = p.H;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    if (p.dq.w_k) dqp_rows<BF16>(p.dq, (int)blockIdx.x, (int)gridDim.x);

    const unsigned int nK = (unsigned)(E / 128);
    unsigned int split_u, tile_u;
    if (!xcd_tile(blockIdx.x, (unsigned)p.splits, nK * nK, split_u, tile_u)) return;
    const int kt_idx = (int)(tile_u % nK), jt_idx = (int)(tile_u / nK);
    const int j0 = jt_idx * 128, k0 = kt_idx * 128;
    const int split = (int)split_u;
    const int64_t rbeg = (int64_t)split * p.rows_per_split;
    const int64_t rend = (rbeg + p.rows_per_split) < p.B ? (rbeg + p.rows_per_split) : p.B;

    const int h_first = j0 / p.hd;
    const int nslots = (j0 + 127) / p.hd - h_first + 1;

    // LDS carve: lhs [2 buffers][hi, lo] | pooled rhs [MAXS slots][hi, lo] | probabilities
    char* ldsL = smem;
    char* ldsR = smem + 4 * HR_TILE;
    f32x2* pl = reinterpret_cast<f32x2*>(ldsR + 2 * MAXS * HR_TILE);

    const int j0w = 32 * (w >> 1), k0w = 64 * (w & 1);
    const int wslot = (j0 + j0w) / p.hd - h_first;
    const bool do_cs = p.colsum != nullptr && kt_idx == 0 && k0w == 0;

    const int q = r16 >> 2, pp = r16 & 3;
    int tx[2][2];
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            tx[b1][hh] = 2048 * lg + 64 * q + 8 * (pp & 1) + 256 * hh + 16 * ((2 * b1 + (pp >> 1)) ^ (2 * (lg & 1) + hh));
    const int a_org = 512 * (w >> 1);
    const int b_org = 1024 * (w & 1);

    f32x4 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csacc[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) csacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* lhs_hi = reinterpret_cast<const char*>(p.lhs);
    const char* lhs_lo = reinterpret_cast<const char*>(p.lhs_lo);
    const char* rhs = reinterpret_cast<const char*>(p.rhs);
    const unsigned int ldl = (unsigned)E * 2u;
    const unsigned int ldr = (unsigned)M_ * (unsigned)E * 2u;

    // pooling role: 8 consecutive lanes = two adjacent rows x four chunks, so that one ds_write_b128 beat covers both 64-byte
    // halves of the 128-byte bank window (row-major roles -- 8 lanes = 8 chunks of one row -- hit the same half twice: a quarter
    // of this kernel's LDS cycles were bank conflicts, profiles/r05_c2_f32params_sq_counters.csv; 158 -> 155 us)
    const int gq = lane >> 3, i8 = lane & 7;
    const int prow = 4 * w + 2 * (gq >> 2) + (i8 >> 2), pc = 4 * (gq & 3) + (i8 & 3);
    const unsigned int prow_off = (unsigned)prow * ldr;
    u32x4 Rb[M_];
    float plr = 0.f;

    auto issue_dma = [&](int64_t base, int buf) {
        const int nvalid = (int)((rend - base) < HRB ? (rend - base) : HRB);
        const int64_t off = base * (int64_t)ldl + (int64_t)j0 * 2;
        dma_hr_images(lhs_hi + off, ldl, nvalid, ldsL + (2 * buf) * HR_TILE);
        dma_hr_images(lhs_lo + off, ldl, nvalid, ldsL + (2 * buf + 1) * HR_TILE);
    };
    const int pt = threadIdx.x / (MAXS * M_), prem = threadIdx.x - pt * (MAXS * M_);
    const int psl = prem / M_, pm = prem - psl * M_;
    auto probs_on = [&](int64_t base) -> bool {
        const int nvalid = (int)((rend - base) < HRB ? (rend - base) : HRB);
        return (int)threadIdx.x < NPL && psl < nslots && pt < nvalid;
    };
    auto load_probs = [&](int64_t base) {
        const float* pu = p.probs + (base * H + h_first) * M_;
        const float* src = pu + (probs_on(base) ? (unsigned)(pt * H * M_ + psl * M_ + pm) : 0u);
        asm volatile("global_load_dword %0, %1, off" : "=v"(plr) : "v"(src) : "memory");
    };
    auto load_x = [&](int64_t base) {
        const int nvalid = (int)((rend - base) < HRB ? (rend - base) : HRB);
        const char* xu = rhs + base * (int64_t)ldr + (int64_t)k0 * 2;
        const unsigned int last = (unsigned)(nvalid - 1) * ldr;
        const unsigned int roff = (prow_off < last ? prow_off : last) + 16u * pc;
#pragma unroll
        for (int m = 0; m < M_; ++m) {
            const char* xm = xu + (size_t)m * E * 2;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(Rb[m]) : "v"(roff), "s"(xm) : "memory");
        }
    };

    if (rbeg < rend) {
        issue_dma(rbeg, 0);
        load_probs(rbeg);
        load_x(rbeg);
    }
    int cur = 0;
    for (int64_t base = rbeg; base < rend; base += HRB, cur ^= 1) {
        const bool more = base + HRB < rend;
        const int nvalid_cur = (int)((rend - base) < HRB ? (rend - base) : HRB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(plr));
        {
            const float pv = probs_on(base) ? plr : 0.f;
            if ((int)threadIdx.x < NPL) pl[threadIdx.x] = f32x2{pv, pv};
        }
#pragma unroll
        for (int m = 0; m < M_; ++m) asm volatile("" : "+v"(Rb[m]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // MFMAs of the previous step done; lhs tiles + probabilities visible
        if (more) {
            issue_dma(base + HRB, cur ^ 1);
            load_probs(base + HRB);
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
        if (more) load_x(base + HRB);
        const int woff = tr_off(prow, pc);
#pragma unroll
        for (int sl = 0; sl < MAXS; ++sl) {
            if (MAXS > 2 && sl >= nslots) continue;       // (two slots: straight-line code, a missing slot has zero probabilities)
            f32x2 pv[4];
            const f32x2 p0 = plc[sl * M_];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = xv[0][i] * p0;
#pragma unroll
            for (int m = 1; m < M_; ++m) {
                const f32x2 pm_ = plc[sl * M_ + m];
#pragma unroll
                for (int i = 0; i < 4; ++i) pv[i] = __builtin_elementwise_fma(xv[m][i], pm_, pv[i]);
            }
            u32x4 oh, ol;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned int h_, l_;
                split_pack(pv[i][0], pv[i][1], h_, l_);
                oh[i] = h_;
                ol[i] = l_;
            }
            char* dst = ldsR + 2 * sl * HR_TILE + woff;
            *reinterpret_cast<u32x4*>(dst) = oh;
            *reinterpret_cast<u32x4*>(dst + HR_TILE) = ol;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // pooled tiles visible
        {
            const char* lt = ldsL + 2 * cur * HR_TILE + a_org;
            const char* rt_tile = ldsR + 2 * wslot * HR_TILE + b_org;
            u32x4 ah[RT], al[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                ah[rt] = tr_frag(lt, tx[rt][0], tx[rt][1]);
                al[rt] = tr_frag(lt + HR_TILE, tx[rt][0], tx[rt][1]);
            }
            if (do_cs) {
                u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                if (nvalid_cur < HRB) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int kk = 8 * lg + 2 * d;
                        ones[d] = (kk < nvalid_cur ? 0x3f80u : 0u) | (kk + 1 < nvalid_cur ? 0x3f800000u : 0u);
                    }
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    csacc[rt] = X::mma(ah[rt], ones, csacc[rt]);
                    csacc[rt] = X::mma(al[rt], ones, csacc[rt]);
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const char* bt = rt_tile + 512 * (ct >> 1);
                const u32x4 bh = tr_frag(bt, tx[ct & 1][0], tx[ct & 1][1]);
                const u32x4 bl = tr_frag(bt + HR_TILE, tx[ct & 1][0], tx[ct & 1][1]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt][ct] = X::mma(ah[rt], bl, acc[rt][ct]);
                    acc[rt][ct] = X::mma(al[rt], bh, acc[rt][ct]);
                    acc[rt][ct] = X::mma(ah[rt], bh, acc[rt][ct]);
                }
            }
        }
    }

    float* out = p.out + (int64_t)split * E * E;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(int64_t)(j0 + j0w + 16 * rt + 4 * lg + r) * E + k0 + k0w + 16 * ct + r16] = acc[rt][ct][r];
    if (do_cs && r16 == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p.colsum[(int64_t)split * E + j0 + j0w + 16 * rt + 4 * lg + r] = csacc[rt][r];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Plain product (dW_o = dy^T o): 512 threads = 8 waves as 2 (j) x 4 (k), block tile 128 x 128, wave tile 64 x 32; every tile by
// LDS-DMA into a RING of D stages, the copy of step k + D - 1 issued at step k: a 32-row step is ~0.2 us of MFMAs, far less
// than a trip to HBM, so one step of lead (the two-buffer form) left every step waiting for its tiles (measured: 64 steps of
// ~1.3 us).  HILO: rhs = o_hi and o_lo (two tiles per stage, two MFMAs per fragment pair); its lhs dy is an exact input.
template <bool HILO, int D>
__global__ __launch_bounds__(512, 4) void gemm_tn_ring_kernel(GemmTnArgs p) {
    using X = Tr<BF16>;
    constexpr int RT = 4, CT = 2, WK = 4;
    constexpr int NI = HILO ? 3 : 2;                                  // DMA instructions per thread and step
    constexpr int STAGE = NI * HR_TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [D stages][lhs | rhs hi | rhs lo]

    const int E = p.E;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    const unsigned int nK = (unsigned)(E / 128);
    unsigned int split_u, tile_u;
    if (!xcd_tile(blockIdx.x, (unsigned)p.splits, nK * nK, split_u, tile_u)) return;
    const int kt_idx = (int)(tile_u % nK), jt_idx = (int)(tile_u / nK);
    const int j0 = jt_idx * 128, k0 = kt_idx * 128;
    const int split = (int)split_u;
    const int64_t rbeg = (int64_t)split * p.rows_per_split;
    const int64_t rend = (rbeg + p.rows_per_split) < p.B ? (rbeg + p.rows_per_split) : p.B;

    const int j0w = 16 * RT * (w / WK), k0w = 16 * CT * (w % WK);
    const bool do_cs = p.colsum != nullptr && kt_idx == 0 && k0w == 0;
    const int q = r16 >> 2, pp = r16 & 3;
    int tx[2][2];
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            tx[b1][hh] = 2048 * lg + 64 * q + 8 * (pp & 1) + 256 * hh + 16 * ((2 * b1 + (pp >> 1)) ^ (2 * (lg & 1) + hh));
    const int a_org = 512 * (j0w >> 5);                   // 16-column block j0w / 16 + rt: pair (j0w >> 5) + (rt >> 1), parity rt & 1
    const int b_org = 512 * (k0w >> 5);

    f32x4 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csacc[RT];
#pragma unroll
    for (int a = 0; a < RT; ++a) csacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* lhs = reinterpret_cast<const char*>(p.lhs);
    const char* rhs_hi = reinterpret_cast<const char*>(p.rhs);
    const char* rhs_lo = reinterpret_cast<const char*>(p.rhs_lo);
    const unsigned int ld = (unsigned)E * 2u;

    auto issue_dma = [&](int64_t base, int stage) {
        const int nvalid = (int)((rend - base) < HRB ? (rend - base) : HRB);
        char* st = smem + stage * STAGE;
        dma_hr_images(lhs + base * (int64_t)ld + (int64_t)j0 * 2, ld, nvalid, st);
        dma_hr_images(rhs_hi + base * (int64_t)ld + (int64_t)k0 * 2, ld, nvalid, st + HR_TILE);
        if (HILO) dma_hr_images(rhs_lo + base * (int64_t)ld + (int64_t)k0 * 2, ld, nvalid, st + 2 * HR_TILE);
    };

#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (rbeg + (int64_t)d * HRB < rend) issue_dma(rbeg + (int64_t)d * HRB, d);
    int cur = 0;
    for (int64_t base = rbeg; base < rend; base += HRB, cur = cur == D - 1 ? 0 : cur + 1) {
        const int nvalid_cur = (int)((rend - base) < HRB ? (rend - base) : HRB);
        // this step's tiles landed: behind them only the copies of the next D - 2 steps may still fly (in-order completion);
        // near the end fewer are in flight than that count assumes, so the tail waits for everything
        if (base + (int64_t)(D - 2) * HRB < rend) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 2) * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // ... for every wave; the stage read at the previous step is free
        char* st = smem + cur * STAGE;
        if (nvalid_cur < HRB) {                                // ragged last step: zero the rhs rows that do not exist
            for (int c = threadIdx.x; c < (HRB - nvalid_cur) * 16; c += 512) {
                const int off = tr_off(nvalid_cur + (c >> 4), c & 15);
                *reinterpret_cast<u32x4*>(st + HR_TILE + off) = u32x4{0u, 0u, 0u, 0u};
                if (HILO) *reinterpret_cast<u32x4*>(st + 2 * HR_TILE + off) = u32x4{0u, 0u, 0u, 0u};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (base + (int64_t)(D - 1) * HRB < rend) issue_dma(base + (int64_t)(D - 1) * HRB, cur == 0 ? D - 1 : cur - 1);
        u32x4 a[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = tr_frag(st + a_org + 512 * (rt >> 1), tx[rt & 1][0], tx[rt & 1][1]);
        if (do_cs) {
            u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
            if (nvalid_cur < HRB) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int kk = 8 * lg + 2 * d;
                    ones[d] = (kk < nvalid_cur ? 0x3f80u : 0u) | (kk + 1 < nvalid_cur ? 0x3f800000u : 0u);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) csacc[rt] = X::mma(a[rt], ones, csacc[rt]);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const char* bt = st + HR_TILE + b_org + 512 * (ct >> 1);
            const u32x4 bh = tr_frag(bt, tx[ct & 1][0], tx[ct & 1][1]);
            if (HILO) {
                const u32x4 bl = tr_frag(bt + HR_TILE, tx[ct & 1][0], tx[ct & 1][1]);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = X::mma(a[rt], bl, acc[rt][ct]);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = X::mma(a[rt], bh, acc[rt][ct]);
        }
    }

    float* out = p.out + (int64_t)split * E * E;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(int64_t)(j0 + j0w + 16 * rt + 4 * lg + r) * E + k0 + k0w + 16 * ct + r16] = acc[rt][ct][r];
    if (do_cs && r16 == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p.colsum[(int64_t)split * E + j0 + j0w + 16 * rt + 4 * lg + r] = csacc[rt][r];
    }
}

template <int M_, int MAXS>
void launch_hilo_pooled(const GemmTnArgs& a, hipStream_t s) {
    const size_t smem = (size_t)(4 + 2 * MAXS) * HR_TILE + (size_t)HRB * MAXS * M_ * 2 * sizeof(float);
    dim3 grid(xcd_grid((unsigned)a.splits, (unsigned)((a.E / 128) * (a.E / 128)))), block(512);
    auto kern = gemm_tn_hilo_pooled_kernel<M_, MAXS>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

int hilo_slots_128(int E, int hd) {               // head slots of the widest aligned 128-row window: 1 or 2 -> 2, 3 or 4 -> 4
    int mx = 1;
    for (int j0 = 0; j0 < E; j0 += 128) {
        const int n = (j0 + 127) / hd - j0 / hd + 1;
        if (n > mx) mx = n;
    }
    return mx <= 2 ? 2 : (mx <= 4 ? 4 : 0);
}

}  // namespace

// shapes of the one-launch hi + lo products: E a multiple of 128 (128 x 128 block tiles), at most 4 head slots per 128 rows,
// M <= 3 (the 128-VGPR budget of the pooled kernel), rows_per_split a multiple of 32
bool gemm_tn_hilo_supported(const GemmTnArgs& a) {               // (the plain product needs rhs_lo at launch, the pooled one lhs_lo)
    if (a.E % 128 != 0 || a.M < 1 || a.M > 3 || (a.Ej > 0 && a.Ej != a.E) || a.rows_per_split % HRB != 0) return false;
    return !a.pooled || hilo_slots_128(a.E, a.hd) != 0;
}

void launch_gemm_tn_hilo(const GemmTnArgs& a, hipStream_t s) {
    if (!a.pooled) {
        // the plain product with both rhs tiles of a step (o_hi, o_lo): a ring of three 24 KB stages, two blocks per CU (75 us at
        // the headline shape; two stages: 85).  (The same kernel without the low tile -- the default product on 32-row steps with
        // four stages -- measured 53 us against the 46 us of the default path's 64-row two-buffer kernel: not instantiated.)
        dim3 grid(xcd_grid((unsigned)a.splits, (unsigned)((a.E / 128) * (a.E / 128)))), block(512);
        const size_t smem = (size_t)3 * 3 * HR_TILE;
        auto kern = gemm_tn_ring_kernel<true, 3>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        kern<<<grid, block, smem, s>>>(a);
        return;
    }
    const bool two128 = hilo_slots_128(a.E, a.hd) == 2;
    switch (a.M) {
        case 1: if (two128) launch_hilo_pooled<1, 2>(a, s); else launch_hilo_pooled<1, 4>(a, s); return;
        case 2: if (two128) launch_hilo_pooled<2, 2>(a, s); else launch_hilo_pooled<2, 4>(a, s); return;
        default: if (two128) launch_hilo_pooled<3, 2>(a, s); else launch_hilo_pooled<3, 4>(a, s); return;
    }
}

}  // namespace aecf
