// Host-callable launchers of the device kernels (internal to libaecf_hip; the public C ABI is
// include/aecf_hip.h).  Every launcher only enqueues work on `s`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aecf_common.h"

namespace aecf {

constexpr int HPAD = 16;   // heads padded to one MFMA column tile

// ---------------- forward ----------------
// qs[j] = (W_q q + b_q)[j] * scale
void launch_prep_qs(int dtype, const void* w_in, const void* b_in, const void* query, float* qs, int E, float scale,
                    hipStream_t s);
// A[h][k] = sum_{j in head h} qs[j] W_k[j][k]   (f32 [HPAD][E] + dtype hi/lo copies for the MFMA B operand)
void launch_prep_amat(int dtype, const void* w_in, const float* qs, float* a_f32, void* a_hi, void* a_lo, int E, int H,
                      hipStream_t s);
// dst[k][j] = src[j][k]   (E x E, dtype)
void launch_transpose(int dtype, const void* src, void* dst, int E, hipStream_t s);
// up to four fragment-major copies of E x E bf16 weight matrices (optionally of the transpose) for the weight-stationary
// kernels' prologue (aecf_gemm_ws.hip): done by the same prep launch
struct FragJobs {
    int n = 0;
    const void* src[4] = {nullptr, nullptr, nullptr, nullptr};
    void* dst[4] = {nullptr, nullptr, nullptr, nullptr};
    int transposed[4] = {0, 0, 0, 0};
};
// qs, folded key matrix (f32 + hi/lo), up to two E x E transposes (t_src* may be null) and the fragment-major copies
// in one launch
void launch_prep_all(int dtype, const void* w_in, const void* b_in, const void* query, float scale, float* qs, float* a_f32,
                     void* a_hi, void* a_lo, const void* t_src0, void* t_dst0, const void* t_src1, void* t_dst1, int E,
                     int H, const FragJobs& fj, hipStream_t s);

struct GateArgs {
    const void* x;            // [B,M,E]
    const void* a_hi;         // [HPAD,E] dtype
    const void* a_lo;         // [HPAD,E] dtype (bf16 only)
    const uint8_t* kpm;       // [B,M] or null
    const float* uniforms;    // [B,M] or null
    float* probs;             // [B,H,M]
    float* attn_w;            // [B,M]
    float* masked_w;          // [B,M] or null
    float* entropy;           // [B] or null
    float* mask_rate;         // [B] or null
    void* i_attn_w;           // the same four in the activation dtype (info copies), each may be null
    void* i_masked_w;
    void* i_entropy;
    void* i_mask_rate;
    void* i_target = nullptr; // [B] activation dtype filled with target_value (info['target_entropy']) or null
    float target_value = 0.f;
    float* ent_partial = nullptr;   // [(B + 255) / 256] sums of (nan_to_num(H) - target_value)^2 per 256 rows, or null (gate_stats only)
    PhiloxDraw ph;                  // mask_mode 1 without a uniforms tensor: the kernel draws them itself (aecf_common.h)
    int64_t B;
    int M, E, H;
    MaskCfg mask;
};
void launch_philox_uniforms(int64_t n, const PhiloxDraw& ph, float* out, hipStream_t s);   // out[i] = philox_uniform_at(ph, i)
void launch_gate_fwd(int dtype, const GateArgs& a, hipStream_t s);
void launch_gate_stats(int dtype, const GateArgs& a, hipStream_t s);   // head mean + masking from probs (scores fused elsewhere)

// C[r][n] = sum_k A(r,k) W[n][k] + bias[n]
//   pooled == 0: A(r,k) = a[r*lda + k]
//   pooled == 1: A(r,k) = sum_m probs[r][head(n)][m] * x[r][m][k]   (a = x [R,M,K], head(n) = n / hd)
struct GemmNtArgs {
    const void* a;
    const void* w;       // [N,K] dtype
    const void* bias;    // [N] dtype or null
    void* c;             // [R,N] dtype
    const float* probs;  // [R,H,M] (pooled only)
    int64_t R;
    int N, K;
    int64_t lda;
    int M, H, hd;
    int pooled;
    int out_f32;         // store C as float32 regardless of dtype (logits)
    void* v_out;         // pooled only: [R,M,N] dtype, the per-modality products W x_m + bias (or null)
    // pooled only, optional: produce the softmax weights in the same kernel (scores x . A[h] against the folded key
    // matrix) instead of reading them; probs is then an OUTPUT.  g_ahi == null: off.
    const void* w_frag = nullptr;     // optional fragment-major copy of w (see FragJobs): faster weight prologue
    const void* g_ahi = nullptr;      // [HPAD,K] dtype
    const void* g_alo = nullptr;      // [HPAD,K] dtype
    const uint8_t* g_kpm = nullptr;   // [R,M] or null
    // plain product only, optional side job of the launch's first block (weight-stationary kernel): the entropy regulariser's
    // final sum -- loss = max(sum(ent_partial[0 .. ent_nblk)) * ent_inv_n, 0) in the activation dtype (ref :309-314) -- so that
    // the forward needs no launch of its own for it (the partials come from the statistics kernel that ran before)
    const float* ent_partial = nullptr;
    int ent_nblk = 0;
    float ent_inv_n = 0.f;
    void* ent_loss = nullptr;         // [1] dtype
    // weight-stationary kernel only, optional: the bf16 LOW part of the output, c_lo = bf16(C - float(bf16(C))), same shape as c
    // (AECF_HILO_GRADS: the weight-gradient products then run on hi + lo operand pairs)
    void* c_lo = nullptr;
};
void launch_gemm_nt(int dtype, const GemmNtArgs& a, hipStream_t s);
void launch_vproj(int dtype, const GemmNtArgs& a, hipStream_t s);   // pooled == 1 path
bool gemm_ws_supported(const GemmNtArgs& a);                          // weight-stationary streaming form (bf16)
void launch_gemm_ws(const GemmNtArgs& a, hipStream_t s);

// fused forward (aecf_row_fwd.hip): prep products in g (a_hi / a_lo), value projection operands in v (w, bias, c = saved o,
// v_out = saved V or null), out-projection operands in y (w, bias, c = y)
bool row_fwd_supported(int dtype, int E, int M, int H);
void launch_row_fwd(const GateArgs& g, const GemmNtArgs& v, const GemmNtArgs& y, hipStream_t s);

// ---------------- backward ----------------
// g_h[b] = W_v,h^T do_h[b] kernels (aecf_bwd_g.hip):
//   dx == false: da[b,h,m] = g_h[b] . x[b,m] -> ds = softmax-backward(da + dwbar/H) -> dsbuf [B,H,M]
//   dx == true : dx[b,m,k] = sum_h probs[b,h,m] g_h[b][k] + sum_h ds[b,h,m] A[h][k]
struct BwdGArgs {
    const void* x;           // [B,M,E]
    const void* dobuf;       // [B,E] dtype  (dy W_o)
    const void* wvt;         // [E(k), E(j)] dtype = W_v^T
    const float* probs;      // [B,H,M]
    const float* d_attn_w;   // [B,M] or null
    const float* d_entropy;  // [B] or null (eval mode)
    const float* attn_w;     // [B,M] (with d_entropy)
    float* dsbuf;            // [B,H,M]  (written by the da pass, read by the dx pass)
    const float* a_f32;      // [HPAD,E]
    void* dx;                // [B,M,E] dtype
    int64_t B;
    int M, E, H, hd;
    float log_M;
    const void* wvt_frag = nullptr;  // optional fragment-major copy of W_v^T (FragJobs): faster dx weight prologue
    int cu_budget = 0;               // dx_ws2: blocks to launch at most (0 = one per CU, 256); fewer leaves CUs to a collective
    // dx_ws2 side job (optional): u_out[H, E] = sum over the u_nslab slabs [H, E] the score-gradient kernel wrote before this
    // launch -- spread over the launch's blocks, done while their weights load -- so that the finalize launch finds u reduced
    const float* u_slab_in = nullptr;
    float* u_out = nullptr;
    int u_nslab = 0;
    // AECF_HILO_GRADS (dsu_ws_kernel): the low part of dobuf -- the score gradient then forms P from do_hi + do_lo, so that the
    // key-side reduction u (dW_q, dW_k, db_q, dquery hang on it) is float32-accurate as well
    const void* do_lo = nullptr;
};
void launch_bwd_g(int dtype, const BwdGArgs& a, bool dx, hipStream_t s);
// score gradient from the saved value projections: da[b,h,m] = do_h[b] . V_h[b,m]  (memory-bound, one wave per sample)
// returns false when the head size is not supported by this kernel (caller falls back to launch_bwd_g(dx = false))
bool launch_dscore_v(int dtype, const BwdGArgs& a, const void* saved_v, hipStream_t s);
bool launch_dx_ws(const BwdGArgs& a, hipStream_t s);
// score gradient + u = ds^T x straight from x (no saved V), bf16 weight-stationary engine with a head split
// (aecf_gemm_ws.hip: dsu_ws_kernel).  dsu_ws_chunks: number of [H, E] u slabs it will write (0 = shape not taken)
int dsu_ws_chunks(const BwdGArgs& a);
int launch_dsu_ws(const BwdGArgs& a, float* u_slab, hipStream_t s);        // bf16 dx on the weight-stationary engine (aecf_gemm_ws.hip)

// out[split][j][k] = sum_{b in split} lhs[b][j] * rhs(b,k)        (f32 partial slabs, deterministic)
//   pooled == 0: rhs(b,k) = rhs[b*E + k]
//   pooled == 1: rhs(b,k) = sum_m probs[b][head(j)][m] x[b][m][k]
// colsum[split][j] = sum_b lhs[b][j]
// pooled == 1 additionally: u[split][h][k] = sum_{b,m} ds[b,h,m] x[b,m,k]
struct GemmTnArgs {
    const void* lhs;      // [B,E] dtype
    const void* rhs;      // [B,E] dtype or x [B,M,E]
    const float* probs;   // pooled
    const float* dsbuf;   // pooled
    float* out;           // [S,E,E]
    float* colsum;        // [S,E]
    float* u;             // [S,HPAD,E] pooled
    int64_t B;
    int M, E, H, hd;
    int Ej;               // lhs feature count when it differs from E (rectangular, non-pooled); 0 = E
    int splits;           // S
    int64_t rows_per_split;   // multiple of 32
    int u_splits;             // the u kernel has its own (finer) batch split: tiny output, needs more blocks
    int64_t u_rows_per_split;
    int pooled;
    int parts = 0;            // pooled: 0 = main product + u, 1 = main product only, 2 = u only (separate stage timing)
    DqpJob dq;                // bf16 transposed-read kernels only: side job of the launch (aecf_common.h); w_k == null: off
    // AECF_HILO_GRADS (aecf_gemm_tn_hilo.hip): the low parts of the operands that are derived values -- lhs_lo = do_lo (pooled
    // product; the pooled rows are split where they are formed), rhs_lo = o_lo (plain product; its lhs dy is an exact input)
    const void* lhs_lo = nullptr;
    const void* rhs_lo = nullptr;
};
void launch_gemm_tn(int dtype, const GemmTnArgs& a, hipStream_t s);
void launch_gemm_tn_tr(const GemmTnArgs& a, hipStream_t s);   // bf16, ds_read_b64_tr_b16 form (main product only)
bool gemm_tn_hilo_supported(const GemmTnArgs& a);             // hi + lo operand pairs in ONE launch (lhs_lo / rhs_lo set)
void launch_gemm_tn_hilo(const GemmTnArgs& a, hipStream_t s);
bool u_mfma_supported(const GemmTnArgs& a);                    // u = ds^T x on the matrix pipe (bf16, H <= 8, E % 128 == 0)
void launch_u_mfma(const GemmTnArgs& a, hipStream_t s);

// dst[g][i] = sum_s src[g][s*n[g] + i] for each of N segments, one launch
struct ReduceSegs {
    static constexpr int N = 5;
    const float* src[N];
    void* dst[N];
    int64_t n[N];
    int splits[N];
    int dst_bf16[N];      // 1: dst is bf16 (one rounding of the float32 sum)
    float scale[N] = {1.f, 1.f, 1.f, 1.f, 1.f};   // the sum is multiplied by this before it is stored (aecf_pool_bwd_args.grad_scale)
};
// tokens of the debug knob AECF_DEBUG (aecf_capi.hip)
bool env_no_ws();
bool env_no_wide_tn();
bool env_no_slab();

void launch_reduce_segments(const ReduceSegs& r, hipStream_t s);

// dW_k[j][k] = qs[j] u[h(j)][k];  dqp[j] = scale * sum_k W_k[j][k] u[h(j)][k]
// dW_q[j][k] = dqp[j] q[k]; db_q = dqp; db_k = 0; dquery[k] = sum_j dqp[j] W_q[j][k]
struct FinalizeArgs {
    const void* w_in;
    const void* query;
    const float* qs;
    const float* u;       // [HPAD,E] reduced
    const float* dqp;     // [E] dq' (DqpJob: side job of the dW_v launch, or launch_dqp)
    float* dq_part;       // (unused since round 4)
    void* dw_in;          // [3E,E]  float32, or bf16 when grad_bf16
    void* db_in;          // [3E]
    void* dquery;         // [E]
    int E, H, hd;
    float scale;
    int grad_bf16;
    float gscale = 1.f;   // every gradient this launch writes is multiplied by it (aecf_pool_bwd_args.grad_scale)
};
void launch_dqp(int dtype, const DqpJob& q, hipStream_t s);        // dq' as its own small launch (shapes without the side job)
void launch_finalize_all(int dtype, const FinalizeArgs& a, const ReduceSegs& r, hipStream_t s);   // slab reduction + the above, ONE launch

// ---------------- stand-alone pieces ----------------
void launch_mask_fwd(int64_t rows, int L, const MaskCfg& cfg, const float* w, const float* u, float* masked,
                     float* entropy, float* mask_rate, uint8_t* bits, hipStream_t s);
void launch_mask_bwd(int64_t rows, int L, int mode, float eps, float log_L, const float* w, const uint8_t* bits,
                     const float* d_masked, const float* d_entropy, float* d_w, hipStream_t s);
// partial[i] = sum over rows 256 i .. of (nan_to_num(H) - target)^2 (one block per 256 rows); loss = max(sum partial, 0) / n
void launch_entropy_partials(int dtype, int64_t n, float target, const void* entropy, float* partial, hipStream_t s);
void launch_entropy_from_partials(int dtype, int64_t n, const float* partial, void* loss, hipStream_t s);
void launch_entropy_loss(int dtype, int64_t n, float target, const void* entropy, float upstream, void* loss,
                         float* d_entropy, float* partial, hipStream_t s);
void launch_sdpa_fwd(int dtype, int64_t B, int S, int T, int E, float scale, const void* q, const void* k, const void* v,
                     void* out, float* probs, hipStream_t s);
void launch_sdpa_bwd(int dtype, int64_t B, int S, int T, int E, float scale, const void* q, const void* k, const void* v,
                     const float* probs, const void* dout, void* dq, void* dk, void* dv, hipStream_t s);

// ---------------- contrastive (aecf_contrastive.hip) ----------------
void launch_l2norm_fwd(int dtype, int64_t n, int d, float eps, const void* z, void* zn, float* inv_norm, hipStream_t s);
void launch_l2norm_bwd(int dtype, int64_t n, int d, const void* zn, const float* inv_norm, const float* dzn, void* dz,
                       hipStream_t s);
// per local row i: logits = S[i,:] * inv_temp; loss_rows[i] = logsumexp - logits[row_offset + i];
// G[i,:] = (softmax - onehot) * coef * inv_temp   (dtype)
void launch_nce_rows(int dtype, int64_t rows, int64_t cols, int64_t row_offset, float inv_temp, float coef, const float* S,
                     void* G, float* loss_rows, hipStream_t s);
// dst[c][r] = src[r][c]   (R x C, dtype; R, C multiples of 32)
void launch_modality_frontend(int dtype, int64_t rows, int dim, const void* feat, const uint8_t* drop, void* out,
                              uint8_t* present, hipStream_t s);
void launch_transpose_rect(int dtype, const void* src, void* dst, int64_t R, int64_t C, hipStream_t s);
void launch_cast_bf16_f32(const void* src, float* dst, int64_t n, hipStream_t s);     // 16-byte aligned src / dst
int launch_cast_f32_bf16_multi(int n, const float* const* src, void* const* dst, const int64_t* numel, hipStream_t s);   // n <= 8

// flash-style InfoNCE direction (aecf_nce_flash.hip): no [rows, cols] logits; optional entropy regulariser in the same call
bool nce_flash_supported(int dtype, int d);
size_t nce_flash_workspace_bytes(int64_t rows, int64_t cols, int d);
void launch_nce_flash(int64_t rows, int64_t cols, int64_t row_offset, int d, float inv_temp, float coef, const void* q,
                      const void* k, float* loss_rows, float* dq, float* dk, void* workspace, const float* ent, int64_t n_ent,
                      float ent_target, float ent_upstream, float* d_ent, float* ent_loss, hipStream_t s);

// InfoNCE on tile GEMMs (aecf_nce_gemm.hip): bf16 exponentials E [rows, cols] in the workspace, one or both directions
bool nce_gemm_supported(int dtype, int d, float temperature);
size_t nce_gemm_workspace_bytes(int64_t rows, int64_t cols, int d);
void launch_nce_gemm_pass1(int64_t rows, int64_t cols, int d, float inv_temp, const void* a, const void* b, void* workspace,
                           float* col_sums, hipStream_t s);
void launch_nce_gemm_loss(int64_t rows, int64_t cols, int64_t row_offset, int d, float inv_temp, int sym, const void* a,
                          const void* b, const float* col_sums, void* workspace, float* loss_rows, const float* ent, int64_t n_ent,
                          float ent_target, float ent_upstream, float* d_ent, float* ent_loss, hipStream_t s);
void launch_nce_gemm_grads(int64_t rows, int64_t cols, int64_t row_offset, int d, float inv_temp, float coef, int sym, const void* a,
                           const void* b, void* workspace, const float* upstream, int out_bf16, void* da, void* db, hipStream_t s);

// ---------------- presence routing (aecf_route.hip) ----------------
void launch_route_build(int64_t rows, const uint8_t* pa, const uint8_t* pb, int32_t* route, int32_t* slot, int32_t* index,
                        int32_t* counts, hipStream_t s);
void launch_rows_gather(int njobs, const void* const* src, const int64_t* src_pitch, const int32_t* const* index,
                        const int64_t* n, void* const* dst, const int64_t* dst_pitch, int64_t row_bytes, hipStream_t s);
void launch_adamw_multi(int n, float* const* p, const float* const* g, float* const* m, float* const* v, float* const* step,
                        const int64_t* numel, unsigned int* ticket, float lr, float beta1, float beta2, float eps, float weight_decay,
                        hipStream_t s);
void launch_rows_split(int64_t rows, int64_t row_bytes, const int32_t* route, const void* src, void* const* dst, hipStream_t s);
void launch_front_pair(int dtype, int64_t rows, int dim_a, int dim_b, const void* feat_a, const void* feat_b, const float* uniforms,
                       float missing_prob, const uint8_t* drop_a, const uint8_t* drop_b, void* out_a, void* out_b,
                       uint8_t* present_a, uint8_t* present_b, int32_t* cls, hipStream_t s);
void launch_rows_select(int64_t rows, int64_t row_bytes, const int32_t* route, const int32_t* slot, const void* const* src,
                        const int64_t* src_pitch, void* dst, int64_t dst_pitch, hipStream_t s);

}  // namespace aecf
