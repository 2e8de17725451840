// gemm_tn: out[j][k] = sum_b lhs[b][j] * rhs(b,k) -- batch-reduction GEMM for the parameter gradients (gfx950).
//
// The reduction index (the batch) is the SLOW axis of both row-major operands, so neither can feed an MFMA
// fragment directly.  Each thread loads an NB x NB block (NB batch rows x NB features; 16-byte coalesced
// loads, 8x8 for bf16 / 4x4 for f32), transposes it in registers and writes it to an LDS tile whose rows are
// FEATURES and whose 128-byte K-slice is the batch (aecf_tile.h); from there both operands are ordinary
// fragment reads.  The staging pass is also where the work that differs from a plain GEMM happens:
//   * column sums of lhs (the bias gradients) fall out of the blocks already in registers;
//   * POOLED: rhs(b,k) = sum_m probs[b, head(j), m] * x[b,m,k] is formed per head slot while staging
//     (the V-projection gradient dW_v,h = do_h^T pooled_h), fp32 FMA, one rounding to the MFMA input type;
//   * POOLED, j-tile 0 only: u[h][k] = sum_{b,m} ds[b,h,m] x[b,m,k] (the key-side gradient) by one extra
//     MFMA chain per modality against a raw transposed x tile.
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

template <typename T, int M_, bool POOLED, int WJ>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs p) {
    using X = Tr<T>;
    typedef typename X::frag frag;
    constexpr int NR = TBlk<T>::NR, NF = TBlk<T>::NF;
    constexpr int BBT = TileK<T>::value;                 // batch rows per step (one 128-byte K slice)
    constexpr int BJ = (WJ == 16) ? 64 : 128;
    constexpr int RT = WJ / 16;
    constexpr int CT = (WJ == 64) ? 4 : 8;
    constexpr int FG = 128 / NF;                         // feature groups per 128 features
    constexpr int BG = BBT / NR;                         // batch groups per step
    constexpr int NBLK_R = FG * BG;                      // register blocks in a [128 features][BBT batch] tile
    constexpr int NBLK_L = (BJ / NF) * BG;
    constexpr int MAXS = 4;                              // head slots per block
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int E = p.E, H = p.H;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    const int j0 = blockIdx.y * BJ, k0 = blockIdx.x * 128;
    const int split = blockIdx.z;
    const int64_t rbeg = (int64_t)split * p.rows_per_split;
    const int64_t rend = (rbeg + p.rows_per_split) < p.B ? (rbeg + p.rows_per_split) : p.B;
    const int jrows = (E - j0) >= BJ ? BJ : (E - j0);    // valid j rows / k cols of this block (multiples of 64)
    const int kcols = (E - k0) >= 128 ? 128 : (E - k0);

    const int h_first = POOLED ? j0 / p.hd : 0;
    const int h_last = POOLED ? (j0 + jrows - 1) / p.hd : 0;
    const int nslots = h_last - h_first + 1;
    const bool do_u = POOLED && blockIdx.y == 0;

    // LDS carve
    char* ldsL = smem;
    char* ldsR = ldsL + BJ * TILE_ROW_BYTES;                                   // nslots tiles of [128][128 B]
    char* ldsX = ldsR + (POOLED ? MAXS : 1) * 128 * TILE_ROW_BYTES;            // raw x tile for u (pooled)
    float* pl = reinterpret_cast<float*>(ldsX + (POOLED ? 128 * TILE_ROW_BYTES : 0));   // probs [BBT][MAXS][M]
    float* csl = pl + (POOLED ? BBT * MAXS * M_ : 0);                          // colsum scratch [BG][BJ]

    // wave tile
    const int wj = (WJ == 64) ? (w >> 1) : w;
    const int j0w = WJ * wj;
    const int k0w = (WJ == 64) ? 64 * (w & 1) : 0;
    const bool wave_on = j0w < jrows && k0w < kcols;
    const int wslot = POOLED ? ((j0 + (j0w < jrows ? j0w : 0)) / p.hd - h_first) : 0;
    const int nct = wave_on ? ((kcols - k0w) >= 16 * CT ? CT : (kcols - k0w) / 16) : 0;

    f32x4 acc[RT][CT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 uacc[2];
    uacc[0] = uacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) cs[f] = 0.f;

    const char* lhs = reinterpret_cast<const char*>(p.lhs);
    const char* rhs = reinterpret_cast<const char*>(p.rhs);
    const int64_t ldl = (int64_t)E * X::BYTES;
    const int64_t ldr = (int64_t)(POOLED ? M_ : 1) * E * X::BYTES;

    // this thread's blocks: q = tid (+256): fg = q % FG (feature group), bg = q / FG (batch group = LDS chunk)
    constexpr int NQL = (NBLK_L + 255) / 256, NQR = (NBLK_R + 255) / 256;
    TBlk<T> Lb[NQL];
    TBlk<T> Rb[NQR][POOLED ? M_ : 1];

    auto load_step = [&](int64_t base) {
#pragma unroll
        for (int i = 0; i < NQL; ++i) {
            const int q = threadIdx.x + 256 * i;
            const int fg = q % (BJ / NF), bg = q / (BJ / NF);
            const bool on = q < NBLK_L && NF * fg < jrows;
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const int64_t bb = base + NR * bg + t;
                if (on && bb < rend) Lb[i].load_row(t, lhs + bb * ldl + (int64_t)(j0 + NF * fg) * X::BYTES);
                else Lb[i].zero_row(t);
            }
        }
#pragma unroll
        for (int i = 0; i < NQR; ++i) {
            const int q = threadIdx.x + 256 * i;
            const int fg = q % FG, bg = q / FG;
            const bool on = q < NBLK_R && NF * fg < kcols;
#pragma unroll
            for (int m = 0; m < (POOLED ? M_ : 1); ++m)
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    const int64_t bb = base + NR * bg + t;
                    if (on && bb < rend)
                        Rb[i][m].load_row(t, rhs + bb * ldr + ((int64_t)m * E + k0 + NF * fg) * X::BYTES);
                    else Rb[i][m].zero_row(t);
                }
        }
    };

    load_step(rbeg);
    for (int64_t base = rbeg; base < rend; base += BBT) {
        __syncthreads();                                  // previous step's MFMAs are done with the tiles
        if (POOLED) {                                     // probabilities of this step's batch rows, per head slot
            for (int i = threadIdx.x; i < BBT * nslots * M_; i += 256) {
                const int t = i / (nslots * M_), rem = i - t * (nslots * M_);
                const int s = rem / M_, m = rem - s * M_;
                const int64_t bb = base + t;
                pl[(t * MAXS + s) * M_ + m] = bb < rend ? p.probs[(bb * H + h_first + s) * M_ + m] : 0.f;
            }
        }
        // lhs blocks: column sums + transposed write
#pragma unroll
        for (int i = 0; i < NQL; ++i) {
            const int q = threadIdx.x + 256 * i;
            if (q < NBLK_L) {
                const int fg = q % (BJ / NF), bg = q / (BJ / NF);
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    float v[NF];
                    Lb[i].get_row(t, v);
#pragma unroll
                    for (int f = 0; f < NF; ++f) cs[f] += v[f];
                }
                Lb[i].store_t(ldsL, NF * fg, bg);
            }
        }
        if (!POOLED) {
#pragma unroll
            for (int i = 0; i < NQR; ++i) {
                const int q = threadIdx.x + 256 * i;
                if (q < NBLK_R) Rb[i][0].store_t(ldsR, NF * (q % FG), q / FG);
            }
            __syncthreads();
        } else {
            __syncthreads();                              // pl visible
#pragma unroll
            for (int i = 0; i < NQR; ++i) {
                const int q = threadIdx.x + 256 * i;
                if (q < NBLK_R) {
                    const int fg = q % FG, bg = q / FG;
                    for (int s = 0; s < nslots; ++s) {
                        TBlk<T> P;
#pragma unroll
                        for (int t = 0; t < NR; ++t) {
                            const float* pr = pl + ((NR * bg + t) * MAXS + s) * M_;
                            float pv[NF], xv[NF];
                            Rb[i][0].get_row(t, xv);
                            const float p0 = pr[0];
#pragma unroll
                            for (int f = 0; f < NF; ++f) pv[f] = p0 * xv[f];
#pragma unroll
                            for (int m = 1; m < M_; ++m) {
                                Rb[i][m].get_row(t, xv);
                                const float pm = pr[m];
#pragma unroll
                                for (int f = 0; f < NF; ++f) pv[f] = fmaf(pm, xv[f], pv[f]);
                            }
                            P.set_row(t, pv);
                        }
                        P.store_t(ldsR + s * 128 * TILE_ROW_BYTES, NF * fg, bg);
                    }
                }
            }
            __syncthreads();
        }

        // raw loads of the next step fly during the MFMAs (u needs this step's raw x first)
        if (do_u) {
            // u[h][k] += sum_m ds[.,h,m]^T x_m : one modality at a time through the raw x tile
#pragma unroll
            for (int m = 0; m < M_; ++m) {
#pragma unroll
                for (int i = 0; i < NQR; ++i) {
                    const int q = threadIdx.x + 256 * i;
                    if (q < NBLK_R) Rb[i][POOLED ? m : 0].store_t(ldsX, NF * (q % FG), q / FG);
                }
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float dv[X::EPL], dl[X::EPL];
#pragma unroll
                    for (int e = 0; e < X::EPL; ++e) {
                        const int64_t bb = base + X::EPL * (4 * ks + lg) + e;
                        dv[e] = (bb < rend && r16 < H) ? p.dsbuf[(bb * H + r16) * M_ + m] : 0.f;
                    }
                    frag dhi = X::pack(dv), dlo = dhi;
                    if (X::BYTES == 2) {
                        float hv[X::EPL];
                        X::unpack(dhi, hv);
#pragma unroll
                        for (int e = 0; e < X::EPL; ++e) dl[e] = dv[e] - hv[e];
                        dlo = X::pack(dl);
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int col0 = 32 * w + 16 * c;
                        if (col0 < kcols) {
                            frag xb = lds_frag<T>(ldsX, col0 + r16, 4 * ks + lg);
                            uacc[c] = X::mma(dhi, xb, uacc[c]);
                            if (X::BYTES == 2) uacc[c] = X::mma(dlo, xb, uacc[c]);
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (base + BBT < rend) load_step(base + BBT);
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

    // ---- slab stores: acc[rt][ct][r] = out[j0 + j0w + 16 rt + 4 lg + r][k0 + k0w + 16 ct + r16] ----
    float* out = p.out + (int64_t)split * E * E;
    if (wave_on) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (ct < nct) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[(int64_t)(j0 + j0w + 16 * rt + 4 * lg + r) * E + k0 + k0w + 16 * ct + r16] = acc[rt][ct][r];
                }
    }
    // column sums: thread (fg, bg) holds partial sums of its NB features over its batch rows; fold the 8 bg in order
    if (blockIdx.x == 0 && p.colsum) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NQL; ++i) {
            const int q = threadIdx.x + 256 * i;
            if (NQL == 1 && q < NBLK_L) {
                const int fg = q % (BJ / NF), bg = q / (BJ / NF);
#pragma unroll
                for (int f = 0; f < NF; ++f) csl[bg * BJ + NF * fg + f] = cs[f];
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < jrows; j += 256) {
            float a = 0.f;
#pragma unroll
            for (int bg = 0; bg < BG; ++bg) a += csl[bg * BJ + j];
            p.colsum[(int64_t)split * E + j0 + j] = a;
        }
    }
    if (do_u) {
        float* u = p.u + (int64_t)split * HPAD * E;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col0 = 32 * w + 16 * c;
            if (col0 < kcols) {
#pragma unroll
                for (int r = 0; r < 4; ++r) u[(int64_t)(4 * lg + r) * E + k0 + col0 + r16] = uacc[c][r];
            }
        }
    }
}

template <typename T, int M_, bool POOLED, int WJ>
static void launch_one(const GemmTnArgs& a, hipStream_t s) {
    constexpr int BJ = (WJ == 16) ? 64 : 128;
    constexpr int BBT = TileK<T>::value;
    size_t smem = (size_t)BJ * TILE_ROW_BYTES + (size_t)(POOLED ? 4 : 1) * 128 * TILE_ROW_BYTES;
    if (POOLED) smem += 128 * TILE_ROW_BYTES + (size_t)BBT * 4 * M_ * sizeof(float);
    smem += (size_t)16 * BJ * sizeof(float);
    dim3 grid((a.E + 127) / 128, (a.E + BJ - 1) / BJ, a.splits), block(256);
    auto kern = gemm_tn_kernel<T, M_, POOLED, WJ>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

template <typename T, int M_, bool POOLED>
static void launch_wj(const GemmTnArgs& a, hipStream_t s) {
    if (!POOLED || a.hd % 64 == 0) launch_one<T, M_, POOLED, 64>(a, s);
    else if (a.hd % 32 == 0) launch_one<T, M_, POOLED, 32>(a, s);
    else launch_one<T, M_, POOLED, 16>(a, s);
}

void launch_gemm_tn(int dtype, const GemmTnArgs& a, hipStream_t s) {
    if (!a.pooled) {
        if (dtype == 0) launch_wj<BF16, 1, false>(a, s); else launch_wj<F32, 1, false>(a, s);
        return;
    }
    AECF_DISPATCH_M(a.M, {
        if (dtype == 0) launch_wj<BF16, M_, true>(a, s); else launch_wj<F32, M_, true>(a, s);
    });
}

}  // namespace aecf
