// gemm_nt: C[R,N] = A(R,K) W[N,K]^T + bias -- LDS-tiled MFMA GEMM (gfx950).
//
// Block = 256 threads = 2x2 waves; block tile 128 rows x 128 cols, wave tile 64 x 64 (4x4 MFMA 16x16 tiles,
// 64 accumulator VGPRs).  K advances 128 bytes per tile (64 bf16 / 32 f32).  Both operands are staged
// through LDS from registers: the global loads of tile t+1 are issued before the MFMAs of tile t and written
// to LDS after the next barrier, so their latency hides under the matrix work (one LDS buffer, two barriers
// per tile).  Rows of the weight matrix come from L2 (<= 2 MB, shared by every block); the activation tile
// is read once per column block.
//
// POOLED (the V projection of the fusion pool): the A operand of output head h is
//     pooled_h[b,:] = sum_m probs[b,h,m] * x[b,m,:]
// formed in registers from the M staged x tiles while the fragments are read (fp32 FMA, one rounding to the
// MFMA input type), so V is projected once per sample instead of once per (sample, modality).
//
// bf16 output leaves through LDS as full 256-byte rows; f32 output is stored from the accumulator layout.
#include <stdlib.h>

#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

// WT = 16 x 16 MFMA tiles per wave and dimension: 4 (block tile 128 x 128) or 1 (32 x 32: problems of a few hundred rows --
// the example model's batch of 64 -- would otherwise run on one or two CUs, 26 us for a [64 x 256] . [256 x 256]^T in float32)
template <typename T, int M_, bool POOLED, int WT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNtArgs p) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    constexpr int NA = POOLED ? M_ : 1;          // A-side LDS tiles (one per modality when pooling)
    constexpr int BT = 32 * WT, WTR = 16 * WT;   // block tile and wave tile edge
    constexpr int TILE = BT * TILE_ROW_BYTES;    // 16 KB (WT = 4)
    constexpr int BK = TileK<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsA = smem;
    char* ldsB = smem + NA * TILE;

    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    const int wr = w >> 1, wc = w & 1;
    unsigned int panel, coltile;
    if (!xcd_tile(blockIdx.x, (unsigned)((p.R + BT - 1) / BT), (unsigned)((p.N + BT - 1) / BT), panel, coltile)) return;
    const int64_t r0 = (int64_t)panel * BT;
    const int n0 = coltile * BT;
    const int rows_valid = (p.R - r0) >= BT ? BT : (int)(p.R - r0);
    const int cols_valid = (p.N - n0) >= BT ? BT : (p.N - n0);
    const int K = p.K;
    const int nkt = K / BK;

    const char* a_src = reinterpret_cast<const char*>(p.a) + r0 * p.lda * X::BYTES;
    const char* w_src = reinterpret_cast<const char*>(p.w) + (int64_t)n0 * K * X::BYTES;
    const int64_t lda_bytes = p.lda * X::BYTES;
    const int64_t ldw_bytes = (int64_t)K * X::BYTES;

    // wave-level column bookkeeping
    const int nw0 = n0 + WTR * wc;
    int nct = (p.N - nw0) >= WTR ? WT : ((p.N - nw0) > 0 ? (p.N - nw0) / 16 : 0);
    int head[4];
    float pr[4][4][M_];
    if (POOLED && WT == 4) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            int h = (nw0 + 16 * ct) / p.hd;
            head[ct] = h < p.H ? h : p.H - 1;
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            int64_t row = r0 + 64 * wr + 16 * rt + r16;
            if (row >= p.R) row = p.R - 1;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int m = 0; m < M_; ++m) pr[rt][ct][m] = p.probs[(row * p.H + head[ct]) * M_ + m];
        }
    }

    f32x4 acc[WT][WT];
#pragma unroll
    for (int rt = 0; rt < WT; ++rt)
#pragma unroll
        for (int ct = 0; ct < WT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (!POOLED) {
        // LDS-DMA, two buffers, one barrier per K tile: barrier (tile kt landed, tile kt-1 consumed) -> issue kt+1 -> MFMAs of kt
        dma_tile<BT, 256>(a_src, lda_bytes, rows_valid, ldsA);
        dma_tile<BT, 256>(w_src, ldw_bytes, cols_valid, ldsB);
        for (int kt = 0; kt < nkt; ++kt) {
            char* curA = ldsA + (kt & 1) * 2 * TILE;
            char* curB = ldsB + (kt & 1) * 2 * TILE;
            __syncthreads();
            if (kt + 1 < nkt) {
                const int64_t koff = (int64_t)(kt + 1) * TILE_ROW_BYTES;
                dma_tile<BT, 256>(a_src + koff, lda_bytes, rows_valid, ldsA + ((kt + 1) & 1) * 2 * TILE);
                dma_tile<BT, 256>(w_src + koff, ldw_bytes, cols_valid, ldsB + ((kt + 1) & 1) * 2 * TILE);
            }
            tile_mma<T, WT, WT>(acc, curA, WTR * wr, curB, WTR * wc);
        }
    }
    // ---------------- epilogue ----------------
    const elem* bias = reinterpret_cast<const elem*>(p.bias);
    float bv[WT];
#pragma unroll
    for (int ct = 0; ct < WT; ++ct) {
        const int n = nw0 + 16 * ct + r16;
        bv[ct] = (bias && ct < nct) ? X::to_f32(bias[n]) : 0.f;
    }
    if (X::BYTES == 2 && !p.out_f32 && WT == 4) {
        // accumulators -> LDS as a [128][128] bf16 image (256-byte rows), then full-row 16-byte stores
        __syncthreads();
        char* cl = smem;
#pragma unroll
        for (int rt = 0; rt < WT; ++rt)
#pragma unroll
            for (int ct = 0; ct < WT; ++ct) {
                // lane r16 holds column c for rows 4*lg + 0..3; pair columns (c, c+1) across lanes r16 ^ 1
                float v0 = acc[rt][ct][0] + bv[ct], v1 = acc[rt][ct][1] + bv[ct];
                float v2 = acc[rt][ct][2] + bv[ct], v3 = acc[rt][ct][3] + bv[ct];
                const bool odd = r16 & 1;
                // even lane keeps rows 0,1 and receives the neighbour's rows 0,1; odd lane keeps rows 2,3
                const float send0 = odd ? v0 : v2, send1 = odd ? v1 : v3;
                const float got0 = __shfl_xor(send0, 1, 64), got1 = __shfl_xor(send1, 1, 64);
                const int col = 64 * wc + 16 * ct + (r16 & ~1);
                const int rowb = 64 * wr + 16 * rt + 4 * lg + (odd ? 2 : 0);
                const unsigned int d0 = odd ? pack_bf16x2(got0, v2) : pack_bf16x2(v0, got0);
                const unsigned int d1 = odd ? pack_bf16x2(got1, v3) : pack_bf16x2(v1, got1);
                *reinterpret_cast<unsigned int*>(cl + (rowb + 0) * 256 + col * 2) = d0;
                *reinterpret_cast<unsigned int*>(cl + (rowb + 1) * 256 + col * 2) = d1;
            }
        __syncthreads();
        char* c = reinterpret_cast<char*>(p.c);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = threadIdx.x + 256 * i;          // 2048 chunks of 16 B: row = ch / 16, chunk = ch % 16
            const int row = ch >> 4, cc = ch & 15;
            if (row < rows_valid && cc * 8 < cols_valid)
                *reinterpret_cast<u32x4*>(c + ((r0 + row) * p.N + n0) * 2 + cc * 16) =
                    *reinterpret_cast<const u32x4*>(cl + row * 256 + cc * 16);
        }
    } else {
        elem* c = reinterpret_cast<elem*>(p.c);
#pragma unroll
        for (int ct = 0; ct < WT; ++ct) {
            if (ct < nct) {
                const int n = nw0 + 16 * ct + r16;
#pragma unroll
                for (int rt = 0; rt < WT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = r0 + WTR * wr + 16 * rt + 4 * lg + r;
                        if (row < p.R) {
                            if (p.out_f32) reinterpret_cast<float*>(p.c)[row * p.N + n] = acc[rt][ct][r] + bv[ct];
                            else c[row * p.N + n] = X::from_f32(acc[rt][ct][r] + bv[ct]);
                        }
                    }
            }
        }
    }
}

template <typename T, int M_, bool POOLED, int WT = 4>
static void launch_one(const GemmNtArgs& a, hipStream_t s) {
    constexpr int NA = POOLED ? M_ : 1;
    constexpr int BT = 32 * WT;
    const size_t smem = (size_t)(POOLED ? (NA + 1) : 4) * BT * TILE_ROW_BYTES;
    dim3 grid(xcd_grid((unsigned)((a.R + BT - 1) / BT), (unsigned)((a.N + BT - 1) / BT))), block(256);
    auto kern = gemm_nt_kernel<T, M_, POOLED, WT>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

void launch_gemm_nt(int dtype, const GemmNtArgs& a_in, hipStream_t s) {
    GemmNtArgs a = a_in;
    if (dtype == 0 && !env_no_ws() && gemm_ws_supported(a)) { launch_gemm_ws(a, s); return; }   // aecf_gemm_ws.hip
    if (a.pooled & 1) { launch_vproj(dtype, a, s); return; }     // per-modality accumulators (aecf_vproj.hip)
    // fewer than 64 block tiles of 128 x 128: 32 x 32 tiles instead (16 x the blocks, each 1/16 of the K loop's MFMAs)
    const bool small = ((a.R + 127) / 128) * (int64_t)((a.N + 127) / 128) < 64;
    if (dtype == 0) { if (small) launch_one<BF16, 1, false, 1>(a, s); else launch_one<BF16, 1, false>(a, s); }
    else { if (small) launch_one<F32, 1, false, 1>(a, s); else launch_one<F32, 1, false>(a, s); }
}

}  // namespace aecf
