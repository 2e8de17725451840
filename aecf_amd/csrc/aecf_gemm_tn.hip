// gemm_tn: out[j][k] = sum_b lhs[b][j] * rhs(b,k) -- batch-reduction GEMM for the parameter gradients (gfx950).
//
// The reduction index (the batch) is the SLOW axis of both row-major operands, so neither can feed an MFMA
// fragment directly.  Each thread loads an NR x NF block (4 batch rows x 8 bf16 / 4 f32 features; 16-byte
// coalesced loads), transposes it in registers and writes it to an LDS tile whose rows are
// FEATURES and whose 128-byte K-slice is the batch (aecf_tile.h); from there both operands are ordinary
// fragment reads.  The staging pass is also where the work that differs from a plain GEMM happens:
//   * column sums of lhs (the bias gradients) fall out of the blocks already in registers;
//   * POOLED: rhs(b,k) = sum_m probs[b, head(j), m] * x[b,m,k] is formed per head slot while staging
//     (the V-projection gradient dW_v,h = do_h^T pooled_h), fp32 FMA, one rounding to the MFMA input type;
//   * POOLED: one extra row of blocks (blockIdx.y == number of j tiles) computes the key-side gradient
//     u[h][k] = sum_{b,m} ds[b,h,m] x[b,m,k] from raw transposed x tiles (A operand = ds, built in registers
//     as a bf16 hi/lo pair so the tiny ds values keep ~16 bits).
// Output: float32 partial slabs per batch split (deterministic; reduced by reduce_slabs).
//
// Block = 256 threads.  Block tile BJ x 128 with BJ = 128 (64 when head_dim % 32 != 0); the wave tile is chosen so
// that a wave's j rows lie inside ONE head (template WJ = 64, 32 or 16 j rows per wave).
#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

// ---- NR x NF register block (NR batch rows x NF features, one 16-byte load per row): load, (optionally
//      combine), transpose, write feature-major to LDS.  bf16: 4 x 8, f32: 4 x 4 -> 256 blocks per tile.
template <typename T> struct TBlk;
template <> struct TBlk<BF16> {
    static constexpr int NR = 4, NF = 8;
    u32x4 r[4];                                   // r[t] = 8 features of batch row t
    __device__ __forceinline__ void load_row(int t, const char* p) { r[t] = *reinterpret_cast<const u32x4*>(p); }
    __device__ __forceinline__ void zero_row(int t) { r[t] = u32x4{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ void get_row(int t, float* v) const { Tr<BF16>::unpack(r[t], v); }
    __device__ __forceinline__ void set_row(int t, const float* v) { r[t] = Tr<BF16>::pack(v); }
    // write feature rows frow0..frow0+7; this block's 4 batch entries are the 8-byte half `bgr & 1` of chunk bgr >> 1
    __device__ __forceinline__ void store_t(char* lds, int frow0, int bgr) const {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            u32x2 o;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const unsigned int a = r[2 * d][f >> 1], b = r[2 * d + 1][f >> 1];
                o[d] = (f & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
            }
            *reinterpret_cast<u32x2*>(lds + lds_off(frow0 + f, bgr >> 1) + 8 * (bgr & 1)) = o;
        }
    }
};
template <> struct TBlk<F32> {
    static constexpr int NR = 4, NF = 4;
    f32x4 r[4];
    __device__ __forceinline__ void load_row(int t, const char* p) { r[t] = *reinterpret_cast<const f32x4*>(p); }
    __device__ __forceinline__ void zero_row(int t) { r[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void get_row(int t, float* v) const { v[0] = r[t][0]; v[1] = r[t][1]; v[2] = r[t][2]; v[3] = r[t][3]; }
    __device__ __forceinline__ void set_row(int t, const float* v) { r[t] = f32x4{v[0], v[1], v[2], v[3]}; }
    __device__ __forceinline__ void store_t(char* lds, int frow0, int bgr) const {
#pragma unroll
        for (int f = 0; f < 4; ++f)
            *reinterpret_cast<f32x4*>(lds + lds_off(frow0 + f, bgr)) = f32x4{r[0][f], r[1][f], r[2][f], r[3][f]};
    }
};

template <typename T, int M_, bool POOLED, int WJ, int MAXS>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTnArgs p) {
    using X = Tr<T>;
    typedef typename X::frag frag;
    constexpr int NR = TBlk<T>::NR, NF = TBlk<T>::NF;
    constexpr int BBT = TileK<T>::value;                 // batch rows per step (one 128-byte K slice)
    constexpr int BJ = (WJ == 16) ? 64 : 128;
    constexpr int RT = WJ / 16;
    constexpr int CT = (WJ == 64) ? 4 : 8;
    constexpr int FG = 128 / NF;                         // feature groups per 128 features
    constexpr int BG = BBT / NR;                         // batch groups per step
    constexpr int NBLK = FG * BG;                        // register blocks per [128 features][BBT batch] tile (256)
    // MAXS: head slots per block (compile time: 1, 2 or 4)
    constexpr int PLN = (BBT * MAXS * M_ + 255) / 256;   // probability values staged per thread per step
    static_assert(NBLK == 256, "one register block per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int E = p.E, H = p.H;
    const int EJ = p.Ej > 0 ? p.Ej : p.E;                // lhs features = output rows (E = rhs features = output cols)
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();

    constexpr bool u_block = false;                               // (u has its own kernel below)
    // 1-D grid: the (j,k) tiles of one batch split share its lhs/rhs rows -> keep them on one XCD (aecf_tile.h)
    const unsigned int nK = (unsigned)((E + 127) / 128), nJt = (unsigned)((EJ + BJ - 1) / BJ);
    unsigned int split_u, tile_u;
    if (!xcd_tile(blockIdx.x, (unsigned)p.splits, nK * nJt, split_u, tile_u)) return;
    const int kt_idx = (int)(tile_u % nK), jt_idx = (int)(tile_u / nK);
    const int j0 = jt_idx * BJ, k0 = kt_idx * 128;
    const int split = (int)split_u;
    const int64_t rbeg = (int64_t)split * p.rows_per_split;
    const int64_t rend = (rbeg + p.rows_per_split) < p.B ? (rbeg + p.rows_per_split) : p.B;
    const int jrows = (EJ - j0) >= BJ ? BJ : (EJ - j0);    // valid j rows / k cols of this block (multiples of 64)
    const int kcols = (E - k0) >= 128 ? 128 : (E - k0);

    const int h_first = POOLED ? j0 / p.hd : 0;
    const int h_last = POOLED ? (j0 + jrows - 1) / p.hd : 0;
    const int nslots = h_last - h_first + 1;

    // LDS carve: L tile | R tiles (head slots, or raw x_m tiles in the u block) | probs | colsum scratch
    char* ldsL = smem;
    char* ldsR = ldsL + BJ * TILE_ROW_BYTES;
    float* pl = reinterpret_cast<float*>(ldsR + (POOLED ? MAXS : 1) * 128 * TILE_ROW_BYTES);   // [BBT][MAXS][M]
    float* csl = pl + (POOLED ? BBT * MAXS * M_ : 0);                                         // [BG][BJ]

    // wave tile of the j x k output
    const int wj = (WJ == 64) ? (w >> 1) : w;
    const int j0w = WJ * wj;
    const int k0w = (WJ == 64) ? 64 * (w & 1) : 0;
    const bool wave_on = !u_block && j0w < jrows && k0w < kcols;
    const int wslot = POOLED ? ((j0 + (j0w < jrows ? j0w : 0)) / p.hd - h_first) : 0;
    const int nct = wave_on ? ((kcols - k0w) >= 16 * CT ? CT : (kcols - k0w) / 16) : 0;

    f32x4 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) cs[f] = 0.f;

    const char* lhs = reinterpret_cast<const char*>(p.lhs);
    const char* rhs = reinterpret_cast<const char*>(p.rhs);
    const int64_t ldl = (int64_t)EJ * X::BYTES;
    const int64_t ldr = (int64_t)(POOLED ? M_ : 1) * E * X::BYTES;

    // this thread's register block: feature group fg, batch group bg (both tiles use the same indices)
    const int fgl = threadIdx.x % (BJ / NF), bgl = threadIdx.x / (BJ / NF);     // lhs tile (BJ features)
    const int fgr = threadIdx.x % FG, bgr = threadIdx.x / FG;                   // rhs tile (128 features)
    const bool l_on = !u_block && bgl < BG && NF * fgl < jrows;
    const bool r_on = NF * fgr < kcols;
    TBlk<T> Lb;
    TBlk<T> Rb[POOLED ? M_ : 1];
    float plr[POOLED ? PLN : 1];

    auto load_step = [&](int64_t base) {
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const int64_t bb = base + NR * bgl + t;
            if (l_on && bb < rend) Lb.load_row(t, lhs + bb * ldl + (int64_t)(j0 + NF * fgl) * X::BYTES);
            else Lb.zero_row(t);
        }
#pragma unroll
        for (int m = 0; m < (POOLED ? M_ : 1); ++m)
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const int64_t bb = base + NR * bgr + t;
                if (r_on && bb < rend) Rb[m].load_row(t, rhs + bb * ldr + ((int64_t)m * E + k0 + NF * fgr) * X::BYTES);
                else Rb[m].zero_row(t);
            }
        if (POOLED && !u_block) {
#pragma unroll
            for (int i = 0; i < PLN; ++i) {
                const int idx = threadIdx.x + 256 * i;          // (t, s, m) with MAXS slots per row
                const int t = idx / (MAXS * M_), rem = idx - t * (MAXS * M_);
                const int s = rem / M_, m = rem - s * M_;
                const int64_t bb = base + t;
                plr[i] = (idx < BBT * MAXS * M_ && s < nslots && bb < rend)
                             ? p.probs[(bb * H + h_first + s) * M_ + m] : 0.f;
            }
        }
    };

    load_step(rbeg);
    for (int64_t base = rbeg; base < rend; base += BBT) {
        __syncthreads();                                  // previous step's MFMAs are done with the tiles
        if (!u_block) {
            if (POOLED) {
#pragma unroll
                for (int i = 0; i < PLN; ++i) {
                    const int idx = threadIdx.x + 256 * i;
                    if (idx < BBT * MAXS * M_) pl[idx] = plr[i];
                }
            }
            if (bgl < BG) {                               // lhs block: column sums + transposed write
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    float v[NF];
                    Lb.get_row(t, v);
#pragma unroll
                    for (int f = 0; f < NF; ++f) cs[f] += v[f];
                }
                Lb.store_t(ldsL, NF * fgl, bgl);
            }
            if (!POOLED) {
                Rb[0].store_t(ldsR, NF * fgr, bgr);
            } else {
                __syncthreads();                          // probabilities visible
                TBlk<T> P[MAXS];
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    float xv[M_][NF];                     // one batch row of every modality, unpacked once
#pragma unroll
                    for (int m = 0; m < M_; ++m) Rb[m].get_row(t, xv[m]);
#pragma unroll
                    for (int s = 0; s < MAXS; ++s) {
                        if (s < nslots) {
                            const float* pr = pl + ((NR * bgr + t) * MAXS + s) * M_;
                            float pv[NF];
                            const float p0 = pr[0];
#pragma unroll
                            for (int f = 0; f < NF; ++f) pv[f] = p0 * xv[0][f];
#pragma unroll
                            for (int m = 1; m < M_; ++m) {
                                const float pm = pr[m];
#pragma unroll
                                for (int f = 0; f < NF; ++f) pv[f] = fmaf(pm, xv[m][f], pv[f]);
                            }
                            P[s].set_row(t, pv);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < MAXS; ++s)
                    if (s < nslots) P[s].store_t(ldsR + s * 128 * TILE_ROW_BYTES, NF * fgr, bgr);
            }
            if (base + BBT < rend) load_step(base + BBT);         // next step's loads fly from here on
            __syncthreads();
            if (wave_on) {
                const char* rt_tile = ldsR + wslot * 128 * TILE_ROW_BYTES;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    frag a[RT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) a[rt] = lds_frag<T>(ldsL, j0w + 16 * rt + r16, 4 * ks + lg);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        if (ct < nct) {
                            frag b = lds_frag<T>(rt_tile, k0w + 16 * ct + r16, 4 * ks + lg);
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = X::mma(a[rt], b, acc[rt][ct]);
                        }
                    }
                }
            }
        }
    }


    // ---- slab stores: acc[rt][ct][r] = out[j0 + j0w + 16 rt + 4 lg + r][k0 + k0w + 16 ct + r16] ----
    float* out = p.out + (int64_t)split * EJ * E;
    if (wave_on) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (ct < nct) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (j0 + j0w + 16 * rt + 4 * lg + r < EJ)         // (EJ need not be a multiple of the wave tile)
                            out[(int64_t)(j0 + j0w + 16 * rt + 4 * lg + r) * E + k0 + k0w + 16 * ct + r16] = acc[rt][ct][r];
                }
    }
    // column sums: thread (fg, bg) holds partial sums of its NF features over its batch rows; fold the bg in order
    if (kt_idx == 0 && p.colsum) {
        __syncthreads();
        if (bgl < BG) {
#pragma unroll
            for (int f = 0; f < NF; ++f) csl[bgl * BJ + NF * fgl + f] = cs[f];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < jrows; j += 256) {
            float a = 0.f;
#pragma unroll
            for (int bg = 0; bg < BG; ++bg) a += csl[bg * BJ + j];
            p.colsum[(int64_t)split * EJ + j0 + j] = a;
        }
    }
}

// u = ds^T x as a streaming kernel on the vector ALU (float32 exact): x is read ONCE, 16 bytes per lane; a wave owns a
// 64-lane slice of the row (512 bf16 / 256 f32 columns) and keeps u[8 heads][its columns] in registers; the softmax-
// gradient scalars ds[b, h, m] are wave-uniform (scalar loads).  8 waves per block walk the block's batch split, then
// fold their partial sums through LDS in a fixed order.  grid (column slices, head groups of 8, batch splits).
template <typename T, int M_>
__global__ __launch_bounds__(512, 2) void u_stream_kernel(GemmTnArgs p) {
    using X = Tr<T>;
    constexpr int CH = X::EPL;                         // columns per lane
    constexpr int HG = 8;                              // heads per block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);       // [8 waves][4 heads][64 * CH]
    const int E = p.E, H = p.H;
    const int lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int col = (blockIdx.x * 64 + lane) * CH;
    const bool c_on = col < E;
    const int h0 = blockIdx.y * HG;
    const int nh = (H - h0) < HG ? (H - h0) : HG;
    const int split = blockIdx.z;
    const int64_t rbeg = (int64_t)split * p.u_rows_per_split;
    const int64_t rend = (rbeg + p.u_rows_per_split) < p.B ? (rbeg + p.u_rows_per_split) : p.B;
    const typename X::elem* x = reinterpret_cast<const typename X::elem*>(p.rhs);

    float acc[HG][CH];
#pragma unroll
    for (int h = 0; h < HG; ++h)
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[h][c] = 0.f;

    for (int64_t b0 = rbeg + w; b0 < rend; b0 += 16) {             // two samples (b0, b0 + 8) per iteration
        typename X::frag xr[2][M_];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int64_t b = b0 + 8 * s2;
#pragma unroll
            for (int m = 0; m < M_; ++m)
                xr[s2][m] = (c_on && b < rend) ? X::load(x + (b * M_ + m) * (int64_t)E + col) : X::zero();
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int64_t b = b0 + 8 * s2;
            if (b < rend) {                                         // wave-uniform
                const float* dsp = p.dsbuf + (b * H + h0) * M_;
#pragma unroll
                for (int m = 0; m < M_; ++m) {
                    float xv[CH];
                    X::unpack(xr[s2][m], xv);
#pragma unroll
                    for (int h = 0; h < HG; ++h) {
                        if (h < nh) {
                            const float d = dsp[h * M_ + m];
#pragma unroll
                            for (int c = 0; c < CH; ++c) acc[h][c] = fmaf(d, xv[c], acc[h][c]);
                        }
                    }
                }
            }
        }
    }
    // fold the 8 waves' partials in wave order, 4 heads per pass
    float* u = p.u + (int64_t)split * H * E;             // slab [split][H][E]: only the heads that exist
    constexpr int SL = 64 * CH;                         // columns of the slice
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
            for (int c = 0; c < CH; ++c) red[(w * 4 + hh) * SL + lane * CH + c] = acc[4 * pass + hh][c];
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * SL; i += 512) {
            const int hh = i / SL, cc = i - hh * SL;
            float a = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) a += red[(ww * 4 + hh) * SL + cc];
            const int h = h0 + 4 * pass + hh, k = blockIdx.x * SL + cc;
            if (h < H && k < E) u[(int64_t)h * E + k] = a;
        }
    }
}

template <typename T, int M_>
static void launch_u(const GemmTnArgs& a, hipStream_t s) {
    constexpr int SL = 64 * Tr<T>::EPL;
    const size_t smem = (size_t)8 * 4 * SL * sizeof(float);
    dim3 grid((a.E + SL - 1) / SL, (a.H + 7) / 8, a.u_splits), block(512);
    auto kern = u_stream_kernel<T, M_>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

template <typename T, int M_, bool POOLED, int WJ, int MAXS>
static void launch_one(const GemmTnArgs& a, hipStream_t s) {
    constexpr int BJ = (WJ == 16) ? 64 : 128;
    constexpr int BBT = TileK<T>::value;
    size_t smem = (size_t)BJ * TILE_ROW_BYTES + (size_t)(POOLED ? MAXS : 1) * 128 * TILE_ROW_BYTES;
    if (POOLED) smem += (size_t)BBT * MAXS * M_ * sizeof(float);
    smem += (size_t)16 * BJ * sizeof(float);
    const int nJ = ((a.Ej > 0 ? a.Ej : a.E) + BJ - 1) / BJ;
    dim3 grid(xcd_grid((unsigned)a.splits, (unsigned)(((a.E + 127) / 128) * nJ))), block(256);
    auto kern = gemm_tn_kernel<T, M_, POOLED, WJ, MAXS>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

// head slots a block needs = the most heads any aligned BJ-row window of the E output rows touches
static int max_slots(int E, int hd, int BJ) {
    int mx = 1;
    for (int j0 = 0; j0 < E; j0 += BJ) {
        const int j1 = (j0 + BJ < E ? j0 + BJ : E) - 1;
        const int n = j1 / hd - j0 / hd + 1;
        if (n > mx) mx = n;
    }
    return mx;
}

template <typename T, int M_, bool POOLED, int WJ>
static void launch_slots(const GemmTnArgs& a, hipStream_t s) {
    if (!POOLED) { launch_one<T, M_, POOLED, WJ, 1>(a, s); return; }
    const int ns = max_slots(a.E, a.hd, (WJ == 16) ? 64 : 128);
    if (ns <= 1) launch_one<T, M_, POOLED, WJ, 1>(a, s);
    else if (ns <= 2) launch_one<T, M_, POOLED, WJ, 2>(a, s);
    else launch_one<T, M_, POOLED, WJ, 4>(a, s);
}

template <typename T, int M_, bool POOLED>
static void launch_wj(const GemmTnArgs& a, hipStream_t s) {
    if (!POOLED || a.hd % 64 == 0) launch_slots<T, M_, POOLED, 64>(a, s);
    else if (a.hd % 32 == 0) launch_slots<T, M_, POOLED, 32>(a, s);
    else launch_slots<T, M_, POOLED, 16>(a, s);
}

void launch_gemm_tn(int dtype, const GemmTnArgs& a, hipStream_t s) {
    // bf16: transposed-LDS-read kernel (aecf_gemm_tn_tr.hip); f32: register-transposed staging (this file)
    if (!a.pooled) {
        if (dtype == 0) launch_gemm_tn_tr(a, s); else launch_wj<F32, 1, false>(a, s);
        return;
    }
    const bool do_main = a.parts != 2, do_u = a.parts != 1;
    if (dtype == 0 && do_main) launch_gemm_tn_tr(a, s);
    if (dtype == 0 && do_u && u_mfma_supported(a) && !env_no_ws()) { launch_u_mfma(a, s); return; }
    AECF_DISPATCH_M(a.M, {
        if (dtype == 0) { if (do_u) launch_u<BF16, M_>(a, s); }
        else { if (do_main) launch_wj<F32, M_, true>(a, s); if (do_u) launch_u<F32, M_>(a, s); }
    });
}

}  // namespace aecf
