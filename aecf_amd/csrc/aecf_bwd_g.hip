// Backward "g" kernels of the fusion pool (gfx950): everything that needs g_h[b] = W_v,h^T do_h[b].
//
//   pass DA (scores):  da[b,h,m] = g_h[b] . x[b,m]  ->  ds = a * (dp - sum_m a dp),  dp = da + dwbar/H
//   pass DX         :  dx[b,m,:] = sum_h a[b,h,m] g_h[b] + sum_h ds[b,h,m] A[h]
//
// g_h is an E-vector per (sample, head); it is produced transposed by MFMA -- rows = E index, columns =
// samples -- from LDS tiles of W_v^T (A operand) and do (B operand), lives only in accumulator registers and
// is consumed at once, so neither g nor the V projection is ever written to memory.
//
// Block = 512 threads (8 waves) owns 64 samples.  Loop: kb over blocks of 128 E-rows; wave w owns the 16
// samples of column tile (w & 3) and the 64 E-rows of half (w >> 2): 4 x 1 MFMA tiles = 16 accumulator
// registers, so the per-sample scalars (probabilities, ds) are read once per head for 4 tiles and the da
// reduction over E needs one lane-group reduce per (head, m).  Inner loop over heads; per (kb, head) the
// K dimension is the head's hd columns, staged through LDS in 128-byte slices with the next slice's global
// loads in flight during the MFMAs.  Small per-wave register footprint -> 2 blocks (16 waves) per CU hide the
// staging latency.  In the accumulator layout a lane holds 4 consecutive E positions of one sample, so x is
// read / dx is written with 8-byte (bf16) or 16-byte (f32) accesses.
//
// The da reduction over E is deterministic (no atomics): the two waves that share a sample tile write their
// partials to their own LDS slots, and after the next barrier a fixed thread adds them in order into its own
// (sample, head, m) cell.
#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

constexpr int G_SAMPLES = 128;               // samples per block
constexpr int G_WAVES = 2 * (G_SAMPLES / 16);  // (sample tiles) x (2 row halves)
constexpr int G_THREADS = 64 * G_WAVES;

template <typename T, int M_, bool DX>
__global__ __launch_bounds__(G_THREADS) void bwd_g_kernel(BwdGArgs p) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    constexpr int BK = TileK<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsA = smem;                               // [128][128 B]  W_v^T rows
    char* ldsB = smem + 128 * TILE_ROW_BYTES;        // [64][128 B]   do rows (samples)
    float* fl = reinterpret_cast<float*>(smem + (128 + G_SAMPLES) * TILE_ROW_BYTES);
    const int E = p.E, H = p.H, hd = p.hd;
    const int HM = H * M_;
    // DA: fl = da[64][H][M] | slot[2][2][64][M]        DX: fl = probs[64][H][M] | ds[64][H][M] | av[H][128]
    float* slot = fl + G_SAMPLES * HM;
    float* avl = fl + 2 * G_SAMPLES * HM;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4, w = wave_id();
    const int64_t b0 = (int64_t)blockIdx.x * G_SAMPLES;
    const int rows_b = (p.B - b0) >= G_SAMPLES ? G_SAMPLES : (int)(p.B - b0);

    if (!DX) {
        for (int i = threadIdx.x; i < G_SAMPLES * HM; i += G_THREADS) fl[i] = 0.f;
    } else {
        for (int i = threadIdx.x; i < G_SAMPLES * HM; i += G_THREADS) {
            const bool ok = (i / HM) < rows_b;
            fl[i] = ok ? p.probs[b0 * HM + i] : 0.f;
            fl[G_SAMPLES * HM + i] = ok ? p.dsbuf[b0 * HM + i] : 0.f;
        }
    }

    const int nkt = (hd + BK - 1) / BK;                          // 128-byte K slices per head
    const int nkb = (E + 127) / 128;
    const int nit = H * nkt;
    const char* wvt = reinterpret_cast<const char*>(p.wvt);
    const char* dob = reinterpret_cast<const char*>(p.dobuf) + b0 * E * X::BYTES;
    const int64_t ld_bytes = (int64_t)E * X::BYTES;

    const int ct = w & (G_SAMPLES / 16 - 1), rh = w / (G_SAMPLES / 16);                           // this wave: samples 16ct.., E-row half rh
    const int srow = 16 * ct + r16;                              // sample (within the block) of this lane
    const int64_t bs = b0 + srow;
    const int64_t bcl = bs < p.B ? bs : p.B - 1;

    DirectStage<128, G_THREADS> sa;
    DirectStage<G_SAMPLES, G_THREADS> sb;
    int pending_h = -1;                                          // DA: head whose slots wait to be folded into da

    // DX with a 2-D grid (few samples: launch_one below): one 128-row block of E per blockIdx.y -- the passes over kb share
    // nothing but the per-sample scalars loaded above
    const int kb_begin = (DX && gridDim.y > 1) ? (int)blockIdx.y : 0;
    const int kb_end = (DX && gridDim.y > 1) ? kb_begin + 1 : nkb;
    for (int kb = kb_begin; kb < kb_end; ++kb) {
        const int rows_k = (E - kb * 128) >= 128 ? 128 : (E - kb * 128);
        const bool wave_on = 64 * rh < rows_k;                   // this wave's 64 E-rows exist
        const int krow = kb * 128 + 64 * rh + 4 * lg;            // E index of accumulator row (rt, r): krow + 16 rt + r

        float xr[DX ? 1 : M_][4][4];
        f32x4 dxa[DX ? M_ : 1][4];
        if (!DX) {
            const elem* x = reinterpret_cast<const elem*>(p.x);
#pragma unroll
            for (int m = 0; m < M_; ++m)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    if (wave_on) X::load4(x + (bcl * M_ + m) * (int64_t)E + krow + 16 * rt, xr[m][rt]);
                    else { xr[m][rt][0] = xr[m][rt][1] = xr[m][rt][2] = xr[m][rt][3] = 0.f; }
                }
        } else {
#pragma unroll
            for (int m = 0; m < M_; ++m)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) dxa[m][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        {
            const int cv = (hd * X::BYTES >= 128) ? 8 : (hd * X::BYTES) / 16;
            sa.load(wvt + (int64_t)(kb * 128) * ld_bytes, ld_bytes, rows_k, cv);
            sb.load(dob, ld_bytes, rows_b, cv);
        }
        f32x4 acc[4][1];
        for (int it = 0; it < nit; ++it) {
            const int h = it / nkt, kt = it - h * nkt;
            if (kt == 0) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
            sa.store(ldsA);
            sb.store(ldsB);
            if (DX && it == 0) {                                 // A[h][kb rows] for every head, used by the epilogues
                for (int i = threadIdx.x; i < H * 128; i += G_THREADS) {
                    const int hh = i >> 7, kk = i & 127;
                    avl[i] = kk < rows_k ? p.a_f32[(int64_t)hh * E + kb * 128 + kk] : 0.f;
                }
            }
            if (!DX && pending_h >= 0) {                         // fold the previous head's two row-half partials
                const float* sl = slot + (pending_h & 1) * (2 * G_SAMPLES * M_);
                for (int i = threadIdx.x; i < G_SAMPLES * M_; i += G_THREADS) {
                    const int s = i / M_, m = i - s * M_;
                    fl[s * HM + pending_h * M_ + m] += sl[i] + sl[G_SAMPLES * M_ + i];
                }
                pending_h = -1;
            }
            __syncthreads();
            if (it + 1 < nit) {
                const int h2 = (it + 1) / nkt, kt2 = (it + 1) - h2 * nkt;
                const int64_t coff = ((int64_t)h2 * hd + (int64_t)kt2 * BK) * X::BYTES;
                const int rem = (hd - kt2 * BK) * X::BYTES;
                const int cv = rem >= 128 ? 8 : rem / 16;
                sa.load(wvt + (int64_t)(kb * 128) * ld_bytes + coff, ld_bytes, rows_k, cv);
                sb.load(dob + coff, ld_bytes, rows_b, cv);
            }
            if (wave_on) tile_mma<T, 4, 1>(acc, ldsA, 64 * rh, ldsB, 16 * ct);
            if (kt != nkt - 1) continue;

            // ---- g_h tile complete: acc[rt][0][r] = g_h[sample srow][E index krow + 16 rt + r] ----
            if (!DX) {
                float* sl = slot + (h & 1) * (2 * G_SAMPLES * M_) + rh * G_SAMPLES * M_;
#pragma unroll
                for (int m = 0; m < M_; ++m) {
                    float s = 0.f;
                    if (wave_on) {
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s = fmaf(acc[rt][0][r], xr[m][rt][r], s);
                    }
                    s = reduce_lg(s);
                    if (lg == 0) sl[srow * M_ + m] = s;
                }
                pending_h = h;
            } else if (wave_on) {
                float pm[M_], dm[M_];
#pragma unroll
                for (int m = 0; m < M_; ++m) {
                    pm[m] = fl[srow * HM + h * M_ + m];
                    dm[m] = fl[G_SAMPLES * HM + srow * HM + h * M_ + m];
                }
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(avl + h * 128 + 64 * rh + 16 * rt + 4 * lg);
#pragma unroll
                    for (int m = 0; m < M_; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            dxa[m][rt][r] = fmaf(pm[m], acc[rt][0][r], fmaf(dm[m], av[r], dxa[m][rt][r]));
                }
            }
        }
        if (!DX) {
            // fold the last head of this kb before the slots are reused by the next kb
            __syncthreads();
            const float* sl = slot + (pending_h & 1) * (2 * G_SAMPLES * M_);
            for (int i = threadIdx.x; i < G_SAMPLES * M_; i += G_THREADS) {
                const int s = i / M_, m = i - s * M_;
                fl[s * HM + pending_h * M_ + m] += sl[i] + sl[G_SAMPLES * M_ + i];
            }
            pending_h = -1;
        } else if (wave_on && bs < p.B) {
            elem* dx = reinterpret_cast<elem*>(p.dx);
#pragma unroll
            for (int m = 0; m < M_; ++m)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    float v[4] = {dxa[m][rt][0], dxa[m][rt][1], dxa[m][rt][2], dxa[m][rt][3]};
                    X::store4(dx + (bs * M_ + m) * (int64_t)E + krow + 16 * rt, v);
                }
        }
    }

    if (!DX) {
        // ds[b,h,:] from the accumulated da (softmax backward), one (sample, head) pair per thread-iteration
        __syncthreads();
        const float invH = 1.0f / (float)H;
        for (int i = threadIdx.x; i < G_SAMPLES * H; i += G_THREADS) {
            const int s = i / H, h = i - s * H;
            if (s >= rows_b) continue;
            const int64_t b = b0 + s;
            float pm[M_], dp[M_], dot = 0.f;
            float ge[M_];
#pragma unroll
            for (int m = 0; m < M_; ++m) ge[m] = 0.f;
            if (p.d_entropy) {       // eval mode: the entropy keeps its graph (ref :150-156)
                float wv[M_], hsum = 0.f;
#pragma unroll
                for (int m = 0; m < M_; ++m) { wv[m] = p.attn_w[b * M_ + m]; hsum -= xlogx(wv[m]); }
                const bool live = (hsum >= 0.f) && (hsum <= p.log_M);
                const float de = p.d_entropy[b];
#pragma unroll
                for (int m = 0; m < M_; ++m) ge[m] = live ? -(logf(wv[m]) + 1.0f) * de : 0.f;
            }
#pragma unroll
            for (int m = 0; m < M_; ++m) {
                pm[m] = p.probs[(b * H + h) * M_ + m];
                const float dwb = (p.d_attn_w ? p.d_attn_w[b * M_ + m] : 0.f) + ge[m];
                dp[m] = fl[s * HM + h * M_ + m] + dwb * invH;
                dot = fmaf(pm[m], dp[m], dot);
            }
#pragma unroll
            for (int m = 0; m < M_; ++m) p.dsbuf[(b * H + h) * M_ + m] = pm[m] * (dp[m] - dot);
        }
    }
}

template <typename T, int M_, bool DX>
static void launch_one(const BwdGArgs& a, hipStream_t s) {
    const size_t HM = (size_t)a.H * a.M;
    const size_t floats = DX ? (2 * G_SAMPLES * HM + (size_t)a.H * 128) : (G_SAMPLES * HM + 2 * 2 * G_SAMPLES * (size_t)a.M);
    const size_t smem = (size_t)(128 + G_SAMPLES) * TILE_ROW_BYTES + floats * sizeof(float);
    dim3 grid((unsigned)((a.B + G_SAMPLES - 1) / G_SAMPLES)), block(G_THREADS);
    if (DX && grid.x < 64) grid.y = (unsigned)((a.E + 127) / 128);       // a few hundred samples: spread the E blocks too
    auto kern = bwd_g_kernel<T, M_, DX>;
    if (smem > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    kern<<<grid, block, smem, s>>>(a);
}

void launch_bwd_g(int dtype, const BwdGArgs& a, bool dx, hipStream_t s) {
    AECF_DISPATCH_M(a.M, {
        if (dtype == 0) {
            if (dx) launch_one<BF16, M_, true>(a, s); else launch_one<BF16, M_, false>(a, s);
        } else {
            if (dx) launch_one<F32, M_, true>(a, s); else launch_one<F32, M_, false>(a, s);
        }
    });
}

}  // namespace aecf
