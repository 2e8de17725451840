// dW_v = sum_b do_h[b]^T pooled_h[b],  pooled_h[b,k] = sum_m probs[b,h,m] x[b,m,k]  -- bf16, gfx950, with the POOLING
// on the matrix pipe as well (the "_pm" form of the pooled batch-reduction GEMM of aecf_gemm_tn_tr.hip).
//
// aecf_gemm_tn_tr.hip forms the pooled operand with vector FMAs (unpack x, v_pk_fma against the probabilities, pack,
// ds_write, barrier, transposed read back): per 64-row step that vector phase, the LDS phase and the MFMAs ran back to
// back, none of them above 40 % busy.  Here the x tile is never touched by a vector instruction:
//   * per 16 samples, head and modality m the pooled tile is ONE MFMA  D[s][k] += Pd[s][s'] X_m[s'][k]  with a DIAGONAL
//     first operand built in registers: row s has two non-zero K slots, (hi, lo) bf16 halves of probs[s,h,m] (so the
//     probabilities enter with 16 mantissa bits), and the second operand is one transposed read of the x tile
//     (ds_read_b64_tr_b16) whose two dwords are used twice (slots 0-3 meet the hi parts, slots 4-7 the lo parts);
//   * sample s sits in D row 4 q + r exactly where the 16x16x32 B operand of the main product wants batch rows
//     8 q + 4 hh + r: two pooled tiles (hh = 0, 1), four v_cvt_pk_bf16_f32, and the pooled operand of 32 batch rows
//     is in registers -- it never exists in LDS;
//   * main product  acc[j][k] += do^T[j][b] pooled[b][k]  as before (lhs by transposed reads of the do tile).
// Block = 512 threads = 8 waves as 4 (64 j = one head) x 2 (64 k), block tile 256 x 128, 32 batch rows per step, per
// wave and step 2 M CT pooling MFMAs + RT CT main MFMAs (M = 3: 24 + 16) and 32 transposed reads.  Tiles AND the
// probabilities arrive by LDS-DMA into a ring of three step buffers (do: two [32][128] images, x: one per modality,
// probabilities: [32][4 heads][M] float32), issued two steps ahead and retired with a counted vmcnt; ONE barrier per step.
// Needs head_dim == 64, E % 256 == 0.
//
// MEASURED (C2, same box): 136 us against 91 us for aecf_gemm_tn_tr.hip's wide kernel -- so this form is OPT-IN
// (AECF_PM=1) and the vector-pooled kernel stays the default.  Ablation (us): empty step loop 32, + main MFMAs 21,
// + pooling 58 (24 MFMAs = ~25 us of matrix time; the rest is building the diagonal operands, 12 v_cvt_pk + 24 selects, and
// two v_mov per MFMA for the doubled x dwords), + LDS-DMA 25.  A first form with 16 thin waves (4 per SIMD) was slower
// still: every wave repeats the per-step bookkeeping (~150 instructions against 20 MFMAs), and register loads of the
// probabilities one step ahead exposed the whole memory latency each step (a step is < 1 us here).  What it taught:
// at these step lengths the matrix pipe is not what a kernel waits for; instruction count per step is.
#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

namespace {

constexpr int PMB = 32;                        // batch rows per step
constexpr int PM_IMG = PMB * 256;              // bytes of one [32][128 bf16] image

typedef short pm_v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pm_v4i16 pm_lds_v4i16;

__device__ __forceinline__ u32x2 tr_read(const char* p) {
    const pm_v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pm_lds_v4i16*)p);
    return __builtin_bit_cast(u32x2, v);
}

template <int ND>
__device__ __forceinline__ void wait_vm() {
    if (ND == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (ND == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (ND == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (ND == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (ND == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (ND == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (ND == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int M_>
__global__ __launch_bounds__(512) void gemm_tn_pm_kernel(GemmTnArgs p) {
    using X = Tr<BF16>;
    constexpr int RT = 4, CT = 4;
    constexpr int NW = 8;                      // waves: 4 (64 j = one head) x 2 (64 k)
    constexpr int NIMG = 2 + M_;               // tile images per step buffer: do[:, 0:128], do[:, 128:256], x_0 .. x_{M-1}
    constexpr int NI = NIMG * 8 + 2;           // DMA wave-instructions per step (1 KB each): the images + the probabilities
    constexpr int ND_HI = (NI + NW - 1) / NW;  // per wave: ND_HI for waves < NI % NW (all when that is 0), else ND_HI - 1
    constexpr int PROB_OFF = NIMG * PM_IMG;    // [32 rows][4 heads][M] float32 = 16 M bytes per row, then padding to 2 KB
    constexpr int STEP_BYTES = PROB_OFF + 2048;
    static_assert(ND_HI <= 7, "vmcnt immediates");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int E = p.E, H = p.H;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());

    const unsigned int nK = (unsigned)(E / 128), nJt = (unsigned)(E / 256);
    unsigned int split_u, tile_u;
    if (!xcd_tile(blockIdx.x, (unsigned)p.splits, nK * nJt, split_u, tile_u)) return;
    const int kt_idx = (int)(tile_u % nK), jt_idx = (int)(tile_u / nK);
    const int j0 = jt_idx * 256, k0 = kt_idx * 128;
    const int split = (int)split_u;
    const int64_t rbeg = (int64_t)split * p.rows_per_split;
    const int64_t rend = (rbeg + p.rows_per_split) < p.B ? (rbeg + p.rows_per_split) : p.B;
    const int nsteps = rbeg < rend ? (int)((rend - rbeg + PMB - 1) / PMB) : 0;

    const int wj = w >> 1, wk = w & 1;         // wave tile: j rows 64 wj .. (head j0 / 64 + wj), k columns 64 wk ..
    const int h_first = j0 >> 6;
    const bool do_cs = p.colsum != nullptr && kt_idx == 0;      // column sums of do: wave (wj, wk) takes row tiles 2 wk, 2 wk + 1

    // transposed-read lane offsets inside a [32][128] image (see aecf_gemm_tn_tr.hip): rows 8 lg + 4 hh + (0..3),
    // 16-column block blk -> 512 (blk >> 1) + tx[blk & 1][hh]
    const int q = r16 >> 2, pp = r16 & 3;
    int tx[2][2];
#pragma unroll
    for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            tx[b1][hh] = 2048 * lg + 64 * q + 8 * (pp & 1) + 256 * hh + 16 * ((2 * b1 + (pp >> 1)) ^ (2 * (lg & 1) + hh));
    const int a_org = (wj >> 1) * PM_IMG + 1024 * (wj & 1);       // do image, then column block 4 (wj & 1) + rt
    const int b_org = 2 * PM_IMG + 1024 * wk;                     // x_m image m at + m PM_IMG; column block 4 wk + ct
    // this lane's probabilities in the step buffer: row 8 q + 4 hh + pp, head slot wj, modality m
    const int p_org = PROB_OFF + (8 * q + pp) * (16 * M_) + 4 * wj * M_;

    // diagonal operand: lane (row i = r16, K group lg) is non-zero only when (i >> 2) == lg; its sample's (hi, lo) pair
    // goes to slots (i & 3) and 4 + (i & 3)
    const bool diag_on = q == lg;
    const unsigned int half_mask = (pp & 1) ? 0xffff0000u : 0x0000ffffu;
    const bool d_second = pp >= 2;             // dword 1 (slots 2, 3) instead of dword 0 (slots 0, 1)

    f32x4 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    const char* lhs = reinterpret_cast<const char*>(p.lhs);
    const char* rhs = reinterpret_cast<const char*>(p.rhs);
    const char* prb = reinterpret_cast<const char*>(p.probs);
    const unsigned int ldl = (unsigned)E * 2u;
    const unsigned int ldr = (unsigned)M_ * (unsigned)E * 2u;
    const unsigned int ldp = (unsigned)(H * M_) * 4u;

    // this wave's DMA pieces: t = w + NW i -> image t >> 3, wave-instruction t & 7 of it (chunks 64 (t & 7) + lane);
    // t >= 8 NIMG: the probabilities (chunk c = 64 (t - 8 NIMG) + lane -> row c / M, 16-byte piece c % M)
    const int my_nd = (NI % NW == 0 || w < NI % NW) ? ND_HI : ND_HI - 1;
    const int tc = 64 * (w & 7) + lane;                               // (t & 7 == w for every i: NW == 8)
    const int drow = 8 * (tc >> 7) + ((tc >> 2) & 7);
    const unsigned int dcol = 16u * (unsigned)(4 * ((tc >> 5) & 3) + ((tc & 3) ^ ((drow >> 2) & 3)));
    auto issue_dma = [&](int s) {
        const int64_t base = rbeg + (int64_t)s * PMB;
        const int nvalid = (int)((rend - base) < PMB ? (rend - base) : PMB);
        char* buf = smem + (s % 3) * STEP_BYTES;
#pragma unroll
        for (int i = 0; i < ND_HI; ++i) {
            const int t = w + NW * i;                         // wave-uniform
            if (t < NI) {
                const int img = t >> 3;
                const char* src;
                unsigned int voff;
                // (the probabilities' two pieces follow the images: PROB_OFF + 1024 u is the same expression)
                const unsigned int dst = (unsigned)(size_t)(lds_void_t*)(buf + img * PM_IMG + 1024 * (t & 7));
                if (img < NIMG) {
                    const int rowc = drow < nvalid ? drow : nvalid - 1;
                    if (img < 2) {
                        src = lhs + base * (int64_t)ldl + (int64_t)(j0 + 128 * img) * 2;
                        voff = (unsigned)rowc * ldl + dcol;
                    } else {
                        src = rhs + base * (int64_t)ldr + (int64_t)(img - 2) * E * 2 + (int64_t)k0 * 2;
                        voff = (unsigned)rowc * ldr + dcol;
                    }
                } else {
                    const int c = 64 * (t - 8 * NIMG) + lane;
                    const int cc = c < PMB * M_ ? c : 0;      // lanes past the last piece re-read piece 0 into the padding
                    const int row = cc / M_, ch = cc - row * M_;
                    const int rowc = row < nvalid ? row : nvalid - 1;
                    src = prb + (base * H + h_first) * (int64_t)(M_ * 4);
                    voff = (unsigned)rowc * ldp + 16u * (unsigned)ch;
                }
                // (s_nop 4: src may come straight out of a v_readfirstlane)
                asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(voff), "s"(src), "s"(dst) : "memory", "m0");
            }
        }
    };

    if (nsteps > 0) {
        issue_dma(0);
        if (nsteps > 1) issue_dma(1);
    }
    for (int s = 0; s < nsteps; ++s) {
        const int64_t base = rbeg + (int64_t)s * PMB;
        const int nvalid = (int)((rend - base) < PMB ? (rend - base) : PMB);
        // this wave's pieces of step s landed; those of step s + 1 (issued last) may still fly
        if (s + 1 < nsteps) {
            if (my_nd == ND_HI) wait_vm<ND_HI>(); else wait_vm<ND_HI - 1>();
        } else {
            wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();      // every wave's pieces of step s landed; step s - 1 is read out everywhere
        if (s + 2 < nsteps) issue_dma(s + 2);

        const char* buf = smem + (s % 3) * STEP_BYTES;
        // (hi, lo) halves of this lane's probabilities, in place in their dword, zero off the diagonal / past the tail
        unsigned int dh[2][M_], dl[2][M_];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const bool on = diag_on && (8 * q + 4 * hh + pp) < nvalid;
            const unsigned int mask = on ? half_mask : 0u;
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                const float pv = *reinterpret_cast<const float*>(buf + p_org + hh * (64 * M_) + 4 * m);
                const unsigned int hb = pack_bf16x2(pv, pv);
                const float ph = __uint_as_float(hb & 0xffff0000u);
                const unsigned int lb = pack_bf16x2(pv - ph, pv - ph);
                dh[hh][m] = hb & mask;
                dl[hh][m] = lb & mask;
            }
        }
        // ---- pooled tiles: D[hh][ct][r] = pooled[sample 8 lg + 4 hh + r][k0 + 64 wk + 16 ct + r16] ----
        f32x4 D[2][CT];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) D[hh][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < M_; ++m)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const u32x4 A = u32x4{d_second ? 0u : dh[hh][m], d_second ? dh[hh][m] : 0u,
                                      d_second ? 0u : dl[hh][m], d_second ? dl[hh][m] : 0u};
                u32x2 xr[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xr[ct] = tr_read(buf + b_org + m * PM_IMG + 512 * (ct >> 1) + tx[ct & 1][hh]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    D[hh][ct] = X::mma(A, u32x4{xr[ct][0], xr[ct][1], xr[ct][0], xr[ct][1]}, D[hh][ct]);
            }
        u32x4 pooled[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            pooled[ct] = u32x4{pack_bf16x2(D[0][ct][0], D[0][ct][1]), pack_bf16x2(D[0][ct][2], D[0][ct][3]),
                               pack_bf16x2(D[1][ct][0], D[1][ct][1]), pack_bf16x2(D[1][ct][2], D[1][ct][3])};
        // ---- main product over the step's 32 batch rows ----
        u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
        if (do_cs && nvalid < PMB) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int kk = 8 * lg + 2 * d;
                ones[d] = (kk < nvalid ? 0x3f80u : 0u) | (kk + 1 < nvalid ? 0x3f800000u : 0u);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const char* at = buf + a_org + 512 * (rt >> 1);
            const u32x2 a0 = tr_read(at + tx[rt & 1][0]), a1 = tr_read(at + tx[rt & 1][1]);
            const u32x4 a = u32x4{a0[0], a0[1], a1[0], a1[1]};
            if (do_cs && (rt >> 1) == wk) csacc[rt & 1] = X::mma(a, ones, csacc[rt & 1]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = X::mma(a, pooled[ct], acc[rt][ct]);
        }
    }

    // ---- slab stores: acc[rt][ct][r] = out[j0 + 64 wj + 16 rt + 4 lg + r][k0 + 64 wk + 16 ct + r16] ----
    float* out = p.out + (int64_t)split * E * E;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(int64_t)(j0 + 64 * wj + 16 * rt + 4 * lg + r) * E + k0 + 64 * wk + 16 * ct + r16] = acc[rt][ct][r];
    if (do_cs && r16 == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p.colsum[(int64_t)split * E + j0 + 64 * wj + 16 * (2 * wk + i) + 4 * lg + r] = csacc[i][r];
    }
}
#pragma clang diagnostic pop

template <int M_>
void launch_pm(const GemmTnArgs& a, hipStream_t s) {
    const size_t smem = (size_t)3 * ((2 + M_) * PM_IMG + 2048);
    dim3 grid(xcd_grid((unsigned)a.splits, (unsigned)((a.E / 128) * (a.E / 256)))), block(512);
    auto kern = gemm_tn_pm_kernel<M_>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

}  // namespace

bool gemm_tn_pm_supported(const GemmTnArgs& a) {
    return a.pooled && a.Ej <= 0 && a.E % 256 == 0 && a.hd == 64 && a.M >= 1 && a.M <= 4 && a.rows_per_split % PMB == 0;
}

void launch_gemm_tn_pm(const GemmTnArgs& a, hipStream_t s) {
    switch (a.M) {
        case 1: launch_pm<1>(a, s); return;
        case 2: launch_pm<2>(a, s); return;
        case 3: launch_pm<3>(a, s); return;
        default: launch_pm<4>(a, s); return;
    }
}

}  // namespace aecf
