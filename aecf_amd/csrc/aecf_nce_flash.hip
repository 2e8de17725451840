// Flash-style InfoNCE, bf16, gfx950: one direction of the contrastive term (aecf_nce_fwd_bwd) without ever materialising
// the [rows, cols] logits -- workspace O(rows d) instead of O(rows cols) (config 3: 8192 x 65536 logits = 2.1 GB float32).
//
//   loss_i = logsumexp_j(q_i.k_j / T) - q_i.k_pos(i) / T                     pos(i) = row_offset + i
//   dq_i   = coef/T (sum_j p_ij k_j - k_pos(i))         p_ij = softmax_j      = "attention output with V = K" - k_pos
//   dk_j   = coef/T (sum_i p_ij q_i - [j = pos(i)] q_i)
//
// One kernel, two roles (template MODE).  A block keeps 64 "stationary" rows (4 waves x 16) as MFMA B operands in
// registers and streams tiles of 32 rows of the other matrix through LDS (LDS-DMA, two buffers).  Per 16 x 16 sub-tile:
//   S[a, b]   = streamed_a . stationary_b          (16x16x32 MFMAs over d; accumulator row = a, column = b)
//   P[a, b]   = DQ: exp(S/T - m_b)                  online maximum m_b / sum l_b per stationary row b
//               DK: coef/T (exp(S/T - lse_a) - [b = pos(a)])      lse_a from the DQ pass
//   Out^T[c, b] += streamed^T[c, a] P[a, b]        16x16x16 MFMAs: the accumulator layout of S (4 consecutive a per lane)
//                                                  IS the B-operand layout of that instruction, and streamed^T is a
//                                                  transposed LDS read (ds_read_b64_tr_b16) of the tile already there
// DQ (stationary = local q, streamed = all keys): the key range is split over KS blocks per row block so that the grid
// fills the chip; each writes its partial (m, l, Out) and a small combine kernel merges them, subtracts k_pos, and
// produces loss_rows and lse.  DK (stationary = keys, streamed = local q): one pass, no partials.  No float atomics:
// results are reproducible.  The entropy regulariser (CurriculumMasking.entropy_loss, ref aecf/AECFLayer.py:285-314)
// rides in the combine launch when asked for, so contrastive + entropy loss and their gradients are one call.
#include <math.h>

#include "aecf_kernels.h"
#include "aecf_tile.h"

namespace aecf {

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

enum { NCE_DQ = 0, NCE_DK = 1 };

struct NceFlashArgs {
    const unsigned short* stat;     // stationary rows [ns, d]   (DQ: q, DK: k)
    const unsigned short* strm;     // streamed rows   [nm, d]   (DQ: k, DK: q)
    int64_t ns, nm;
    int64_t row_offset;             // pos(i) = row_offset + i for local row i
    float inv_temp, coef;
    const float* lse;               // DK: [nm] log-sum-exp of the streamed (local q) rows
    float* part_m;                  // DQ: [KS, ns]
    float* part_l;                  // DQ: [KS, ns]
    float* part_o;                  // DQ: [KS, ns, d]
    float* out;                     // DK: dk [ns, d]
    int ksplit;                     // DQ: key splits
    int64_t strm_per_split;         // DQ: streamed rows per split (multiple of 32)
};

// CSPLIT > 1: the output columns are produced in CSPLIT launches of D / CSPLIT columns each (cpart = which), every one
// recomputing S -- the 16 x D float32 accumulator of a wave plus the stationary fragments exceed the register file of a
// wave beyond D = 512 (DQ, whose rescaling touches the accumulator with vector instructions) / D = 768 (DK).
template <int KT, int MODE, int CSPLIT>
__global__ __launch_bounds__(256, 1) void nce_flash_kernel(NceFlashArgs p, int cpart) {
    using X = Tr<BF16>;
    constexpr int D = 32 * KT, NC = D / 16 / CSPLIT, ROWB = 2 * D;
    const int c_first = cpart * NC;
    constexpr int TILE = 32 * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(wave_id());
    const int nsb = (int)((p.ns + 63) / 64);
    const int sb = (int)blockIdx.x % nsb, split = (int)blockIdx.x / nsb;
    const int64_t s0 = (int64_t)sb * 64 + 16 * w;                 // this wave's 16 stationary rows
    const int64_t m_beg = MODE == NCE_DQ ? (int64_t)split * p.strm_per_split : 0;
    const int64_t m_end = MODE == NCE_DQ ? ((m_beg + p.strm_per_split) < p.nm ? (m_beg + p.strm_per_split) : p.nm) : p.nm;
    if (m_beg >= m_end) {                                         // an empty key split (block-uniform): neutral partial
        if (MODE == NCE_DQ && lg == 0 && cpart == 0 && s0 + r16 < p.ns) {
            p.part_m[(int64_t)split * p.ns + s0 + r16] = -INFINITY;
            p.part_l[(int64_t)split * p.ns + s0 + r16] = 0.f;
        }
        return;
    }

    const char* msrc = reinterpret_cast<const char*>(p.strm);
    auto issue = [&](int64_t m0, int buf) {
        const int mv = (int)((m_end - m0) < 32 ? (m_end - m0) : 32);
        ws_dma_rows_asm<KT, 32, 1, 256>(msrc + m0 * (int64_t)ROWB, (unsigned)ROWB, mv, smem + buf * TILE);
    };
    issue(m_beg, 0);

    // stationary rows as B operands: lane (lg, r16 = b): row s0 + r16, elements 32 ks + 8 lg .. + 7
    u32x4 sreg[KT];
    {
        int64_t srow = s0 + r16;
        srow = srow < p.ns ? srow : p.ns - 1;
        const unsigned short* sp = p.stat + srow * D + 8 * lg;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) sreg[ks] = *reinterpret_cast<const u32x4*>(sp + 32 * ks);
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) asm volatile("" : "+v"(sreg[ks]));      // retire the loads before the loop
    }
    f32x4 oacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float run_m = -INFINITY, run_l = 0.f;                         // DQ: per stationary row b = r16 (replicated over lg)
    const int64_t pos_b = MODE == NCE_DK ? (s0 + r16) : 0;        // DK: key index of this lane's column

    // fragment / transposed-read addresses inside a tile (rows a, 16-byte chunk ^ (row & 15))
    int aaddr[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) aaddr[v] = r16 * ROWB + ((((4 * v) + lg) ^ r16) << 4);
    const int q = r16 >> 2, pp = r16 & 3;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int cur = 0;
    for (int64_t m0 = m_beg; m0 < m_end; m0 += 32, cur ^= 1) {
        __builtin_amdgcn_s_barrier();
        if (m0 + 32 < m_end) issue(m0 + 32, cur ^ 1);
        const char* tb = smem + cur * TILE;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int64_t a0 = m0 + 16 * sub;                     // streamed rows a0 .. a0 + 15 of this sub-tile
            if (a0 >= m_end) break;                               // block-uniform
            const char* ts = tb + 16 * sub * ROWB;
            // ---- S[a, b]: A = streamed rows (LDS), B = stationary rows (registers)
            f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                const u32x4 af = *reinterpret_cast<const u32x4*>(ts + aaddr[ks & 3] + (ks >> 2) * 256);
                sacc = X::mma(af, sreg[ks], sacc);
            }
            // lane (lg, r16): S[a = a0 + 4 lg + r][b = s0 + r16], r = 0..3
            float pv[4];
            if (MODE == NCE_DQ) {
                float tmax = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = (a0 + 4 * lg + r < m_end) ? sacc[r] * p.inv_temp : -INFINITY;
                    tmax = fmaxf(tmax, pv[r]);
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float new_m = fmaxf(run_m, tmax);
                if (__any(new_m > run_m)) {                       // wave-uniform: rescale the running sums
                    const float sc = (run_m == -INFINITY) ? 0.f : expf(run_m - new_m);
                    run_l *= sc;
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) oacc[c][r] *= sc;
                    run_m = new_m;
                }
                float ts_ = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { pv[r] = expf(pv[r] - run_m); ts_ += pv[r]; }
                ts_ += __shfl_xor(ts_, 16, 64);
                ts_ += __shfl_xor(ts_, 32, 64);
                run_l += ts_;
            } else {
                const float ct = p.coef * p.inv_temp;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t a = a0 + 4 * lg + r;            // local q row
                    float v = 0.f;
                    if (a < m_end) {
                        v = expf(sacc[r] * p.inv_temp - p.lse[a]);
                        if (p.row_offset + a == pos_b) v -= 1.0f;
                        v *= ct;
                    }
                    pv[r] = v;
                }
            }
            // ---- Out^T[c, b] += streamed^T[c, a] P[a, b]   (16x16x16: B operand = P as it sits in the accumulator)
            const u32x2 pb2 = u32x2{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
            const s16x4 pb = __builtin_bit_cast(s16x4, pb2);
            // A operand: lane (lg, r16 = c): streamed rows 4 lg .. 4 lg + 3 at column 16 ct + r16 -- one transposed read;
            // lane 4 q + pp of the group supplies row 4 lg + q, columns 16 ct + 4 pp .. + 3
            const int trow = 4 * lg + q;
            const int tbase = trow * ROWB + 8 * (pp & 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ch = (2 * (c_first + c) + (pp >> 1)) ^ trow;    // key(row) = row & 15 = trow (16-row sub-tile)
                const v4i16_t at = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)(ts + tbase + (ch << 4)));
                oacc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, at), pb, oacc[c], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- epilogue: lane (lg, r16 = b) holds Out[b][16 c + 4 lg + r]
    const int64_t b = s0 + r16;
    if (b < p.ns) {
        if (MODE == NCE_DQ) {
            float* po = p.part_o + ((int64_t)split * p.ns + b) * D + 16 * c_first + 4 * lg;
#pragma unroll
            for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(po + 16 * c) = oacc[c];
            if (lg == 0 && cpart == 0) {
                p.part_m[(int64_t)split * p.ns + b] = run_m;
                p.part_l[(int64_t)split * p.ns + b] = run_l;
            }
        } else {
            float* po = p.out + b * D + 16 * c_first + 4 * lg;
#pragma unroll
            for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(po + 16 * c) = oacc[c];
        }
    }
}

// merge the key splits of a row: m* = max m_s, l* = sum l_s e^(m_s - m*), O* = sum O_s e^(m_s - m*);
// dq = coef/T (O*/l* - k_pos), lse = m* + log l*, loss = lse - q.k_pos/T.  One wave per row.
// Block 0 additionally reduces the entropy regulariser (optional): loss_e = mean((H - target)^2), d_entropy.
struct NceCombineArgs {
    const unsigned short* q;
    const unsigned short* k;
    const float* part_m;
    const float* part_l;
    const float* part_o;
    float* dq;
    float* loss_rows;
    float* lse;
    int64_t rows, row_offset;
    int d, ksplit;
    float inv_temp, coef;
    // entropy regulariser riding in this launch (n_ent == 0: off)
    const float* ent;
    float* d_ent;
    float* ent_loss;
    int64_t n_ent;
    float ent_target, ent_scale;
};

__global__ __launch_bounds__(256) void nce_combine_kernel(NceCombineArgs p) {
    const int lane = lane_id();
    const int64_t i = (int64_t)blockIdx.x * 4 + wave_id();
    if (i < p.rows) {
        float mstar = -INFINITY;
        for (int s = 0; s < p.ksplit; ++s) mstar = fmaxf(mstar, p.part_m[(int64_t)s * p.rows + i]);
        float lstar = 0.f;
        for (int s = 0; s < p.ksplit; ++s) {
            const float ms = p.part_m[(int64_t)s * p.rows + i];
            lstar += (ms == -INFINITY) ? 0.f : p.part_l[(int64_t)s * p.rows + i] * expf(ms - mstar);
        }
        const unsigned short* kp = p.k + (p.row_offset + i) * p.d;
        const unsigned short* qp = p.q + i * p.d;
        const float ct = p.coef * p.inv_temp, inv_l = 1.0f / lstar;
        float dot = 0.f;
        for (int c = lane; c < p.d; c += 64) {
            float o = 0.f;
            for (int s = 0; s < p.ksplit; ++s) {
                const float ms = p.part_m[(int64_t)s * p.rows + i];
                if (ms != -INFINITY) o += p.part_o[((int64_t)s * p.rows + i) * p.d + c] * expf(ms - mstar);
            }
            const float kv = Tr<BF16>::to_f32(kp[c]);
            p.dq[i * p.d + c] = ct * (o * inv_l - kv);
            dot = fmaf(Tr<BF16>::to_f32(qp[c]), kv, dot);
        }
        dot = reduce_wave(dot);
        if (lane == 0) {
            const float lse = mstar + logf(lstar);
            p.lse[i] = lse;
            p.loss_rows[i] = lse - dot * p.inv_temp;
        }
    }
    if (blockIdx.x == 0 && p.n_ent > 0) {                         // entropy regulariser: one block, fixed order
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

template <int KT, int MODE>
void launch_flash_mode(const NceFlashArgs& a, int blocks, hipStream_t s) {
    constexpr int D = 32 * KT;
    constexpr int CSPLIT = (MODE == NCE_DQ ? (KT >= 32 ? 4 : (KT >= 24 ? 2 : 1)) : (KT >= 32 ? 2 : 1));
    const size_t smem = (size_t)2 * 32 * 2 * D;
    auto kern = nce_flash_kernel<KT, MODE, CSPLIT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int cpart = 0; cpart < CSPLIT; ++cpart) kern<<<dim3((unsigned)blocks), dim3(256), smem, s>>>(a, cpart);
}

}  // namespace

bool nce_flash_supported(int dtype, int d) {
    return dtype == 0 && (d == 128 || d == 256 || d == 384 || d == 512 || d == 768 || d == 1024);
}

// key splits of the DQ pass: enough blocks to fill 256 CUs, at least 512 keys per split
int nce_flash_ksplit(int64_t rows, int64_t cols) {
    const int64_t rb = (rows + 63) / 64;
    int64_t ks = (256 + rb - 1) / rb;
    const int64_t max_ks = (cols + 511) / 512;
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
    if (ks > 64) ks = 64;
    return (int)ks;
}

size_t nce_flash_workspace_bytes(int64_t rows, int64_t cols, int d) {
    const int ks = nce_flash_ksplit(rows, cols);
    return ((size_t)ks * rows * (d + 2) + (size_t)rows) * sizeof(float) + 1024;
}

void launch_nce_flash(int64_t rows, int64_t cols, int64_t row_offset, int d, float inv_temp, float coef, const void* q,
                      const void* k, float* loss_rows, float* dq, float* dk, void* workspace, const float* ent, int64_t n_ent,
                      float ent_target, float ent_upstream, float* d_ent, float* ent_loss, hipStream_t s) {
    const int ks = nce_flash_ksplit(rows, cols);
    float* ws = reinterpret_cast<float*>(workspace);
    float* part_m = ws;
    float* part_l = part_m + (size_t)ks * rows;
    float* part_o = part_l + (size_t)ks * rows;
    float* lse = part_o + (size_t)ks * rows * d;
    NceFlashArgs a;
    a.stat = (const unsigned short*)q; a.strm = (const unsigned short*)k; a.ns = rows; a.nm = cols; a.row_offset = row_offset;
    a.inv_temp = inv_temp; a.coef = coef; a.lse = nullptr; a.part_m = part_m; a.part_l = part_l; a.part_o = part_o;
    a.out = nullptr; a.ksplit = ks;
    a.strm_per_split = ((cols + ks - 1) / ks + 31) / 32 * 32;
    NceFlashArgs b = a;
    b.stat = (const unsigned short*)k; b.strm = (const unsigned short*)q; b.ns = cols; b.nm = rows; b.lse = lse; b.out = dk;
    b.ksplit = 1; b.strm_per_split = rows;
    NceCombineArgs c;
    c.q = (const unsigned short*)q; c.k = (const unsigned short*)k; c.part_m = part_m; c.part_l = part_l; c.part_o = part_o;
    c.dq = dq; c.loss_rows = loss_rows; c.lse = lse; c.rows = rows; c.row_offset = row_offset; c.d = d; c.ksplit = ks;
    c.inv_temp = inv_temp; c.coef = coef; c.ent = ent; c.d_ent = d_ent; c.ent_loss = ent_loss; c.n_ent = ent ? n_ent : 0;
    c.ent_target = ent_target; c.ent_scale = n_ent > 0 ? 2.0f * ent_upstream / (float)n_ent : 0.f;
#define NCE_KT(KT_)                                                                                  \
    launch_flash_mode<KT_, NCE_DQ>(a, (int)(((rows + 63) / 64) * ks), s);                            \
    nce_combine_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>(c);                   \
    launch_flash_mode<KT_, NCE_DK>(b, (int)((cols + 63) / 64), s);
    switch (d / 32) {
        case 4: NCE_KT(4) break;
        case 8: NCE_KT(8) break;
        case 12: NCE_KT(12) break;
        case 16: NCE_KT(16) break;
        case 24: NCE_KT(24) break;
        default: NCE_KT(32) break;
    }
#undef NCE_KT
}

}  // namespace aecf
