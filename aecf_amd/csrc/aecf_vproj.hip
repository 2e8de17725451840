// V projection of the fusion pool, forward (gfx950):
//     o[b, n] = sum_m a[b, head(n), m] * (x[b,m,:] . W_v[n,:]) + b_v[n]
// One MFMA accumulator set PER MODALITY over the same staged W_v tile; the softmax weights are applied to the
// accumulators in the epilogue (M FMAs per output element).  Compared with pooling x before the MFMA this spends
// M x the matrix flops and saves ~7*H*M*E vector instructions per sample -- the vector ALU, not the matrix
// pipe, was the bound (profiles/r01_pmc_notes.md).
//
// Block = 256 threads = 2x2 waves; block tile 64 samples x 128 columns; wave tile 32 x 64 (2 x 4 MFMA tiles per
// modality).  LDS: M x-tiles [64][128 B] + one W tile [128][128 B], register-staged one tile ahead.
#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

// CW = 16-column MFMA tiles per wave: 4 (block tile 64 x 128), or 1 (64 x 32) for a few hundred samples, which would otherwise
// sit on one or two CUs (the example model's batch of 64: 26 us in float32)
template <typename T, int M_, int CW>
__global__ __launch_bounds__(256, 2) void vproj_modal_kernel(GemmNtArgs p) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    constexpr int ATILE = 64 * TILE_ROW_BYTES;    // 8 KB
    constexpr int BK = TileK<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsA = smem;
    char* ldsB = smem + M_ * ATILE;
    constexpr int BC = 32 * CW, WC = 16 * CW;                               // columns per block / per wave
    float* prl = reinterpret_cast<float*>(ldsB + BC * TILE_ROW_BYTES);      // probs [64][H][M]

    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    const int wr = w >> 1, wc = w & 1;
    unsigned int panel, coltile;
    if (!xcd_tile(blockIdx.x, (unsigned)((p.R + 63) / 64), (unsigned)((p.N + BC - 1) / BC), panel, coltile)) return;
    const int64_t r0 = (int64_t)panel * 64;
    const int n0 = coltile * BC;
    const int rows_valid = (p.R - r0) >= 64 ? 64 : (int)(p.R - r0);
    const int cols_valid = (p.N - n0) >= BC ? BC : (p.N - n0);
    const int K = p.K, H = p.H, HM = p.H * M_;
    const int nkt = K / BK;
    const int nw0 = n0 + WC * wc;
    const int nct = (p.N - nw0) >= WC ? CW : ((p.N - nw0) > 0 ? (p.N - nw0) / 16 : 0);

    for (int i = threadIdx.x; i < 64 * HM; i += 256)
        prl[i] = (i / HM) < rows_valid ? p.probs[r0 * HM + i] : 0.f;

    const char* a_src = reinterpret_cast<const char*>(p.a) + r0 * p.lda * X::BYTES;
    const char* w_src = reinterpret_cast<const char*>(p.w) + (int64_t)n0 * K * X::BYTES;
    const int64_t lda_bytes = p.lda * X::BYTES;
    const int64_t ldw_bytes = (int64_t)K * X::BYTES;

    f32x4 acc[M_][2][CW];
#pragma unroll
    for (int m = 0; m < M_; ++m)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < CW; ++ct) acc[m][rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    DirectStage<64, 256> sa[M_];
    DirectStage<BC, 256> sb;
#pragma unroll
    for (int m = 0; m < M_; ++m) sa[m].load(a_src + (int64_t)m * K * X::BYTES, lda_bytes, rows_valid);
    sb.load(w_src, ldw_bytes, cols_valid);

    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M_; ++m) sa[m].store(ldsA + m * ATILE);
        sb.store(ldsB);
        __syncthreads();
        if (kt + 1 < nkt) {
            const int64_t koff = (int64_t)(kt + 1) * TILE_ROW_BYTES;
#pragma unroll
            for (int m = 0; m < M_; ++m) sa[m].load(a_src + (int64_t)m * K * X::BYTES + koff, lda_bytes, rows_valid);
            sb.load(w_src + koff, ldw_bytes, cols_valid);
        }
        if (nct > 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                typename X::frag b[CW];
#pragma unroll
                for (int ct = 0; ct < CW; ++ct) b[ct] = lds_frag<T>(ldsB, WC * wc + 16 * ct + r16, 4 * ks + lg);
#pragma unroll
                for (int m = 0; m < M_; ++m)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const typename X::frag a = lds_frag<T>(ldsA + m * ATILE, 32 * wr + 16 * rt + r16, 4 * ks + lg);
#pragma unroll
                        for (int ct = 0; ct < CW; ++ct) acc[m][rt][ct] = X::mma(a, b[ct], acc[m][rt][ct]);
                    }
            }
        }
    }

    // ---------------- epilogue: weight the per-modality products, add the bias ----------------
    const elem* bias = reinterpret_cast<const elem*>(p.bias);
    f32x4 o[2][CW];
#pragma unroll
    for (int ct = 0; ct < CW; ++ct) {
        const int n = nw0 + 16 * ct + r16;
        const float bv = (bias && ct < nct) ? X::to_f32(bias[n]) : 0.f;
        int h = (nw0 + 16 * ct) / p.hd;
        h = h < H ? h : H - 1;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* pr = prl + (32 * wr + 16 * rt + 4 * lg + r) * HM + h * M_;
                float v = bv;
#pragma unroll
                for (int m = 0; m < M_; ++m) v = fmaf(pr[m], acc[m][rt][ct][r], v);
                o[rt][ct][r] = v;
            }
    }
    // ---- stores.  bf16: ONE LDS pass -- image [64 rows][1 + M slots][128 cols] (slot 0 = o, slot 1+m = V_m, the
    //      per-modality products W x_m + bias kept for the backward score gradient), then full-row 16-byte stores.
    if (X::BYTES == 2 && !p.out_f32 && CW == 4) {
        const int nslot = p.v_out ? M_ + 1 : 1;
        const int pitch = nslot * 256;
        __syncthreads();
        char* cl = smem;
        auto put = [&](int slot, int rt, int ct, float v0, float v1, float v2, float v3) {
            const bool odd = r16 & 1;
            const float send0 = odd ? v0 : v2, send1 = odd ? v1 : v3;
            const float got0 = __shfl_xor(send0, 1, 64), got1 = __shfl_xor(send1, 1, 64);
            const int col = 64 * wc + 16 * ct + (r16 & ~1);
            const int rowb = 32 * wr + 16 * rt + 4 * lg + (odd ? 2 : 0);
            char* base = cl + slot * 256 + col * 2;
            *reinterpret_cast<unsigned int*>(base + (rowb + 0) * pitch) = odd ? pack_bf16x2(got0, v2) : pack_bf16x2(v0, got0);
            *reinterpret_cast<unsigned int*>(base + (rowb + 1) * pitch) = odd ? pack_bf16x2(got1, v3) : pack_bf16x2(v1, got1);
        };
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < CW; ++ct) {
                put(0, rt, ct, o[rt][ct][0], o[rt][ct][1], o[rt][ct][2], o[rt][ct][3]);
                if (p.v_out) {
                    const int n = nw0 + 16 * ct + r16;
                    const float bv = (bias && ct < nct) ? X::to_f32(bias[n]) : 0.f;
#pragma unroll
                    for (int m = 0; m < M_; ++m)
                        put(1 + m, rt, ct, acc[m][rt][ct][0] + bv, acc[m][rt][ct][1] + bv, acc[m][rt][ct][2] + bv,
                            acc[m][rt][ct][3] + bv);
                }
            }
        __syncthreads();
        char* c = reinterpret_cast<char*>(p.c);
        char* vo = reinterpret_cast<char*>(p.v_out);
        for (int i = threadIdx.x; i < 64 * 16 * nslot; i += 256) {      // 16-byte chunks: (row, slot, chunk)
            const int cc = i & 15, slot = (i >> 4) % nslot, row = (i >> 4) / nslot;
            if (row < rows_valid && cc * 8 < cols_valid) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(cl + row * pitch + slot * 256 + cc * 16);
                if (slot == 0) *reinterpret_cast<u32x4*>(c + ((r0 + row) * p.N + n0) * 2 + cc * 16) = v;
                else *reinterpret_cast<u32x4*>(vo + (((r0 + row) * M_ + (slot - 1)) * p.N + n0) * 2 + cc * 16) = v;
            }
        }
    } else {
        elem* c = reinterpret_cast<elem*>(p.c);
        elem* vo = reinterpret_cast<elem*>(p.v_out);
#pragma unroll
        for (int ct = 0; ct < CW; ++ct) {
            if (ct < nct) {
                const int n = nw0 + 16 * ct + r16;
                const float bv = bias ? X::to_f32(bias[n]) : 0.f;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = r0 + 32 * wr + 16 * rt + 4 * lg + r;
                        if (row < p.R) {
                            if (X::BYTES == 2 && p.out_f32) reinterpret_cast<float*>(p.c)[row * p.N + n] = o[rt][ct][r];
                            else c[row * p.N + n] = X::from_f32(o[rt][ct][r]);
                            if (vo) {
#pragma unroll
                                for (int m = 0; m < M_; ++m) vo[(row * M_ + m) * p.N + n] = X::from_f32(acc[m][rt][ct][r] + bv);
                            }
                        }
                    }
            }
        }
    }
}

template <typename T, int M_, int CW>
static void launch_one(const GemmNtArgs& a, hipStream_t s) {
    constexpr int BC = 32 * CW;
    size_t smem = (size_t)M_ * 64 * TILE_ROW_BYTES + (size_t)BC * TILE_ROW_BYTES + (size_t)64 * a.H * M_ * sizeof(float);
    const size_t image = CW == 4 ? (size_t)64 * 256 * (a.v_out ? M_ + 1 : 1) : 0;     // the bf16 output image(s)
    if (smem < image) smem = image;
    dim3 grid(xcd_grid((unsigned)((a.R + 63) / 64), (unsigned)((a.N + BC - 1) / BC))), block(256);
    auto kern = vproj_modal_kernel<T, M_, CW>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

void launch_vproj(int dtype, const GemmNtArgs& a, hipStream_t s) {
    const bool small = ((a.R + 63) / 64) * (int64_t)((a.N + 127) / 128) < 64;
    AECF_DISPATCH_M(a.M, {
        if (dtype == 0) { if (small) launch_one<BF16, M_, 1>(a, s); else launch_one<BF16, M_, 4>(a, s); }
        else { if (small) launch_one<F32, M_, 1>(a, s); else launch_one<F32, M_, 4>(a, s); }
    });
}

}  // namespace aecf
