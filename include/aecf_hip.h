/*
 * aecf_hip.h -- C ABI of libaecf_hip.so: the MI355X (gfx950) implementation of the AECF
 * fusion hot path (BASELINE.json north_star).
 *
 * The reference (leochlon/aecf) is pure Python and has no FFI of its own: its boundary for
 * this path is the Python API in aecf/AECFLayer.py.  Each entry point below replaces the
 * arithmetic of one reference function; the Python package aecf_amd/ binds these symbols
 * with ctypes and mirrors the reference's signatures (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - plain C symbols, plain pointers and sizes, no C++/torch types.
 *  - the CALLER owns every buffer (including the workspace); the library never allocates or
 *    frees device memory, never retains a caller pointer past the call, never synchronises
 *    the stream and never throws.
 *  - every call only enqueues kernels on `stream` (a hipStream_t passed as void*) -- plain launches: a caller that is
 *    capturing `stream` (a whole training step as one HIP graph) gets them as nodes of its own graph.
 *  - return value: AECF_OK (0) or a negative aecf_status.
 *  - dtype: AECF_BF16 or AECF_F32 for x / query / weights / y.  All statistics (attention weights, entropy, mask rate,
 *    saved probabilities) are float32 (optional copies in the activation dtype: aecf_pool_fwd_args.info_*).  Parameter
 *    gradients are float32, or bf16 when aecf_pool_bwd_args.grad_dtype asks for it (bf16 parameters: the float32 batch
 *    sums are rounded once, in the reduction kernel).
 *  - thread-safe and stateless: any thread may call with any stream of the current device; the library keeps nothing between
 *    calls.  The only process-wide input is ONE debug knob, read once: the environment variable AECF_DEBUG = comma-separated
 *    tokens -- no_ws, no_gate_fusion, no_wide_tn, no_slab (route a shape through the kernels that serve the shapes the fast
 *    ones do not take: what the parity tests of those kernels use), fused_fwd (the one-kernel forward instead of the
 *    weight-stationary kernel pair), dx_reserve=N.  Unknown tokens are reported on stderr and ignored; unset = production
 *    behaviour.
 */
#ifndef AECF_HIP_H
#define AECF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AECF_ABI_VERSION 9

typedef enum aecf_status {
    AECF_OK = 0,
    AECF_ERR_BAD_DIMS = -1,        /* non-positive size, E % num_heads != 0, ...            */
    AECF_ERR_UNSUPPORTED = -2,     /* valid for the reference but not built here (see msg)  */
    AECF_ERR_NULL_POINTER = -3,
    AECF_ERR_WORKSPACE = -4,       /* workspace too small                                    */
    AECF_ERR_LAUNCH = -5           /* hipGetLastError() != hipSuccess after a launch         */
} aecf_status;

typedef enum aecf_dtype { AECF_BF16 = 0, AECF_F32 = 1 } aecf_dtype;

/* Problem description shared by forward and backward.
 * Replaces: MultimodalAttentionPool.__init__/forward shape contract
 * (ref aecf/AECFLayer.py:371-407, 461-498) for the hot-path case
 *   query = fusion_query.expand(B,1,E)  (one shared query, ref :694-695),
 *   key is value = x[B,M,E] contiguous, batch_first, dropout 0, no attn_mask. */
typedef struct aecf_pool_desc {
    int64_t batch;        /* B                                   */
    int32_t modalities;   /* M = src_len, 1..8                    */
    int32_t embed_dim;    /* E, multiple of 64                    */
    int32_t num_heads;    /* H, 1..16, E % H == 0, (E/H) % 32 == 0 (bf16) or % 16 == 0 (f32) */
    int32_t dtype;        /* aecf_dtype of x/query/weights/y      */
    /* curriculum masking (ref aecf/AECFLayer.py:76-99); mask_mode 0 = no CurriculumMasking
     * module attached, 1 = training-mode masking (:158-283), 2 = eval-mode (:150-156)      */
    int32_t mask_mode;
    int32_t min_active;
    float base_mask_prob;
    float entropy_target;
    float eps;            /* the module's _eps buffer, 1e-8 (:96) */
} aecf_pool_desc;

/* Forward.  Replaces nn.MultiheadAttention math (torch functional.py:5836-5852, 6576-6612,
 * called at ref :515-521) + CurriculumMasking.forward (ref :130-283) on the head-averaged
 * weights, fused.  Outputs other than y are float32. */
typedef struct aecf_pool_fwd_args {
    const void* x;               /* [B,M,E] dtype                                        */
    const void* query;           /* [E] dtype: the shared fusion query                   */
    const void* w_in;            /* [3E,E] dtype  (in_proj_weight, q|k|v)                */
    const void* b_in;            /* [3E] dtype or NULL                                   */
    const void* w_out;           /* [E,E] dtype   (out_proj.weight)                      */
    const void* b_out;           /* [E] dtype or NULL                                    */
    const uint8_t* key_padding_mask; /* [B,M] nonzero = ignore, or NULL                  */
    const float* uniforms;       /* [B,M] float32 U[0,1) (mask_mode 1) or NULL           */
    void* y;                     /* [B,E] dtype                                          */
    float* attn_w;               /* [B,M] head-averaged weights (info['attention_weights']) */
    float* masked_w;             /* [B,M] or NULL                                        */
    float* entropy;              /* [B]   or NULL                                        */
    float* mask_rate;            /* [B]   or NULL                                        */
    float* saved_probs;          /* [B,H,M] per-head softmax, saved for backward         */
    void* saved_o;               /* [B,E] dtype: pre-out-projection heads, saved for backward (or NULL) */
    void* saved_v;               /* [B,M,E] dtype: per-modality value projections W_v x + b_v, saved so that the
                                  * backward score gradient is a memory-bound dot instead of a recompute (or NULL) */
    void* workspace;
    size_t workspace_bytes;
    /* optional profiling hook: AECF_FWD_STAGES+1 hipEvent_t handles (caller-created); event[0] is recorded
     * before the first kernel and event[i] after stage i (see aecf_pool_stage_name). NULL = off. */
    void** stage_events;
    /* optional copies of the four info tensors in the ACTIVATION dtype, written by the same kernel (the reference
     * returns info in the input dtype, ref :520-543, so a bf16 caller needs no cast kernels); NULL = not wanted */
    void* info_attn_w;           /* [B,M] dtype */
    void* info_masked_w;         /* [B,M] dtype */
    void* info_entropy;          /* [B]   dtype */
    void* info_mask_rate;        /* [B]   dtype */
    /* optional: aecf_pool_prep_bytes(desc) bytes that receive everything the BACKWARD derives from the parameters alone
     * (scaled query projection, folded key matrix, W_v^T / W_o^T and their MFMA-fragment copies), produced by the
     * forward's own preparation launch.  Hand the same buffer to aecf_pool_backward (parameters unchanged in
     * between -- what autograd guarantees) and its preparation stage disappears.  NULL = off. */
    void* saved_prep;
    /* optional (mask_mode 1): info['target_entropy'] of the reference (ref :273, full_like(entropy, log(M) * target)):
     * [B] dtype filled with target_entropy_value by the kernel that writes the other info tensors; NULL = not wanted */
    void* info_target_entropy;
    float target_entropy_value;  /* the caller's log(M) * entropy_target, rounded to float32 */
    /* AECF_PRECISE (desc.dtype == AECF_BF16 only): the float32-STORE form of the bf16 path ("bf16
     * tolerance"): x / query / weights are bf16 as usual and every product of exact bf16 operands runs on the bf16 MFMA
     * kernels, but no intermediate is rounded to bf16 -- the pooled heads o are kept in float32 and the products that
     * consume an intermediate accumulate from it in float32.  y, saved_o (and in the backward dx and all parameter
     * gradients) are then float32 buffers; workspace = aecf_pool_precise_workspace_bytes.  A verification mode (it moves
     * twice the bytes): what "outputs match within 1e-3 relative, bf16" is asserted on. */
    int32_t flags;
    /* optional (mask_mode 1; ABI v7): partial sums of CurriculumMasking.entropy_loss over the rows of this call (ref
     * aecf/AECFLayer.py:285-314), float32 [(B + 255) / 256]: entry i = sum over rows 256 i .. of (nan_to_num(H) - target)^2 with
     * H = the entropy as the info tensor holds it and target = target_entropy_value.  Written by the kernel that writes the
     * entropies, so that a later entropy_loss of exactly this tensor is ONE small launch (aecf_entropy_loss_from_partials)
     * instead of a pass over the rows.  NULL = not wanted. */
    float* ent_loss_partial;
    /* ABI v8.  AECF_DRAW_UNIFORMS (mask_mode 1, uniforms == NULL): the statistics kernel draws the Bernoulli uniforms of ref
     * aecf/AECFLayer.py:204 itself -- for weight element i exactly the value `torch.rand(B*M, device=...)` would hold at i when
     * the device generator stands at (philox_seed, philox_offset): Philox4x32-10, subsequence = the element's thread of
     * torch's launch of philox_threads threads (its grid: min(ceil(n / 256), CUs * 8) blocks of 256), 2^-32 + v 2^-32 with 1.0
     * mapped to 0.0.  The caller advances its generator by what that torch call would have consumed (4 per 4*threads elements).
     * One launch and a [B,M] float32 tensor less per step; a caller who needs a recorded draw (parity tests, data-parallel
     * shards of one global draw) keeps passing `uniforms`. */
    uint64_t philox_seed;
    uint64_t philox_offset;
    uint32_t philox_threads;
    /* ABI v8, optional (mask_mode 1, with ent_loss_partial): [1] dtype, CurriculumMasking.entropy_loss over the rows of this
     * call -- max(mean((nan_to_num(H) - target)^2), 0), ref :285-314 -- written by the out-projection launch's first block from
     * the partial sums (no launch of its own; shapes the weight-stationary kernel does not take: one small launch).  NULL = off. */
    void* ent_loss;
    /* ABI v8, AECF_HILO_GRADS: [B,E] dtype, the LOW part of saved_o -- bf16(o - float(bf16(o))) -- written beside it by the
     * value projection; hand both to the backward (see aecf_pool_bwd_args.saved_o_lo).  Required with the flag. */
    void* saved_o_lo;
    /* ABI v9, AECF_DRAW_UNIFORMS: this call's rows are rows [r0, r0 + B) of a LARGER batch whose mask uniforms are ONE draw
     * `torch.rand(B_global * M)` -- a data-parallel rank's shard.  philox_element0 = r0 * M is the element of that draw the
     * call's first weight element is, and philox_threads is the thread count of the GLOBAL launch; weight element i of the call
     * takes the value the global tensor holds at philox_element0 + i.  Every rank passes the same (seed, offset) and advances its
     * generator by what the global call consumes: N-rank masks are the one-rank masks bit for bit with no uniforms tensor and no
     * launch.  0 = the call is the whole draw. */
    int64_t philox_element0;
} aecf_pool_fwd_args;

#define AECF_PRECISE 1
#define AECF_DRAW_UNIFORMS 2
/* AECF_HILO_GRADS (bf16; aecf_pool_hilo_bwd_workspace_bytes(desc) > 0 says the shape is built): the WEIGHT-GRADIENT products
 * of the backward take the operands that are derived values as bf16 hi + lo pairs -- dW_o = dy^T (o_hi + o_lo), dW_v = do_hi^T
 * P_hi + do_hi^T P_lo + do_lo^T P_hi with do = dy W_o and the pooled rows P split where they are formed, the key-side reduction
 * from do_hi + do_lo -- so that float32-STORED parameter gradients are float32-accurate (the bf16 roundings of the derived
 * operands are what puts the default path's dW at 1.3 - 2.3e-3 of fp32 math at the headline shape: above north_star's 1e-3).
 * Everything else (y, dx, the kernels) is the default path.  Set it on BOTH calls.  ONE launch per product (round 5,
 * aecf_gemm_tn_hilo.hip: hi and lo tiles of a step side by side in LDS, one accumulator set, one slab set); the backward
 * workspace is the larger aecf_pool_hilo_bwd_workspace_bytes (do_lo).  The Python layer sets it by itself whenever the parameter
 * gradients are float32-stored (float32 master weights under bf16 activations); with bf16 parameters the gradient's own
 * rounding (2^-9) hides what the flag buys. */
#define AECF_HILO_GRADS 4
/* AECF_PREP_READY (forward only, with saved_prep; ABI v9): saved_prep already holds the preparation an earlier forward made from
 * these very parameters and this query (same embed_dim / num_heads / dtype; the batch may differ): the preparation launch is
 * skipped.  For loops in which the parameters provably stand still -- inference, gradient accumulation over micro-batches.  The
 * caller is responsible for "unchanged" (the Python layer keys its cache on the parameters' version counters and storage, in
 * eval mode / without gradient recording only). */
#define AECF_PREP_READY 8

/* Backward (autograd transpose of the above). */
typedef struct aecf_pool_bwd_args {
    const void* x;
    const void* query;
    const void* w_in;
    const void* b_in;
    const void* w_out;
    const void* dy;              /* [B,E] dtype                                          */
    const float* d_attn_w;       /* [B,M] grad on info['attention_weights'] or NULL      */
    const float* d_entropy;      /* [B] grad on info['entropy'] (eval mode only) or NULL */
    const float* attn_w;         /* [B,M] forward output (needed with d_entropy)         */
    const float* saved_probs;    /* [B,H,M]                                              */
    const void* saved_o;         /* [B,E]                                                */
    const void* saved_v;         /* [B,M,E] or NULL (NULL: the score gradient recomputes W_v^T do per head)  */
    void* dx;                    /* [B,M,E] dtype                                        */
    void* dquery;                /* [E]     grad_dtype                                   */
    void* dw_in;                 /* [3E,E]  grad_dtype                                   */
    void* db_in;                 /* [3E]    grad_dtype                                   */
    void* dw_out;                /* [E,E]   grad_dtype                                   */
    void* db_out;                /* [E]     grad_dtype                                   */
    void* workspace;
    size_t workspace_bytes;
    void** stage_events;         /* AECF_BWD_STAGES+1 hipEvent_t handles or NULL (profiling hook) */
    /* element type of the five parameter gradients: AECF_F32, or AECF_BF16 when desc.dtype is AECF_BF16 (the
     * float32 batch sums are rounded once, in the reduction kernel -- what autograd's cast to a bf16 parameter does) */
    int32_t grad_dtype;
    int32_t flags;               /* AECF_PRECISE: dy bf16; saved_o, dx and the gradients float32 (see aecf_pool_fwd_args.flags) */
    const void* saved_prep;      /* buffer filled by aecf_pool_forward (see aecf_pool_fwd_args.saved_prep) or NULL */
    /* optional hipEvent_t (caller-created): recorded on `stream` as soon as ALL FIVE parameter gradients (dquery, dw_in,
     * db_in, dw_out, db_out) are final.  With it set the backward computes the input gradient dx LAST (otherwise it comes
     * before dW_v), so a data-parallel caller can run the gradients' all-reduce on another stream behind the dx kernel
     * (aecf_amd/dp.py: GradOverlap).  stage_events are not recorded in this order of stages.
     * NULL = off. */
    void* param_grads_event;
    const void* saved_o_lo;      /* AECF_HILO_GRADS: the forward's low part of saved_o, else NULL (ABI v8) */
    /* ABI v9: the five parameter gradients are multiplied by grad_scale as their float32 batch sums are stored (before the one
     * rounding of a bf16 gradient) -- a data-parallel caller passes 1 / world, so that the gradient average is ONE sum
     * all-reduce with no divide launch around it.  dx is not scaled.  0 = 1 (off). */
    float grad_scale;
} aecf_pool_bwd_args;

#define AECF_FWD_STAGES 4   /* prep, gate, vproj, outproj */
#define AECF_BWD_STAGES 8   /* prep, dout, dw_out, dscore, u, dx, dw_v, finalize */

int aecf_abi_version(void);
/* name of forward (backward == 0) or backward stage i, or NULL when out of range */
const char* aecf_pool_stage_name(int backward, int stage);
const char* aecf_status_string(int status);

/* validates the description; returns AECF_OK or the reason it cannot run */
int aecf_pool_check(const aecf_pool_desc* d);
/* scratch bytes needed by forward / backward for this description */
size_t aecf_pool_fwd_workspace_bytes(const aecf_pool_desc* d);
size_t aecf_pool_bwd_workspace_bytes(const aecf_pool_desc* d);
/* 1 when the backward of this description is faster with the forward's per-modality value projections
 * (aecf_pool_fwd_args.saved_v), 0 when it derives the score gradient from x itself and saved_v should stay NULL
 * (bf16, E in {256, 512}, M <= 4: the forward then writes B*M*E fewer elements) */
int aecf_pool_wants_saved_v(const aecf_pool_desc* d);
/* workspace bytes of a call with AECF_PRECISE set (backward == 0: forward) */
size_t aecf_pool_precise_workspace_bytes(const aecf_pool_desc* d, int backward);
/* backward workspace bytes of a call with AECF_HILO_GRADS set; 0 = the flag is not built for this description */
size_t aecf_pool_hilo_bwd_workspace_bytes(const aecf_pool_desc* d);
/* bytes of the optional parameter-preparation buffer shared by forward and backward (saved_prep) */
size_t aecf_pool_prep_bytes(const aecf_pool_desc* d);

/* The generator call of AECF_DRAW_UNIFORMS on its own (tests, callers that want the tensor): out[i], i < n, = the float32
 * uniform described at aecf_pool_fwd_args.philox_seed for element element0 + i of the draw (element0: see philox_element0).  aecf_philox_host evaluates one element on the host (known-answer
 * tests; `raw` != NULL additionally receives the four 32-bit outputs of the Philox4x32-10 block the element comes from). */
int aecf_philox_uniforms(int64_t n, uint64_t seed, uint64_t offset, uint32_t threads, int64_t element0, float* out, void* stream);
float aecf_philox_host(uint64_t seed, uint64_t offset, uint32_t threads, int64_t element, uint32_t* raw);

int aecf_pool_forward(const aecf_pool_desc* d, const aecf_pool_fwd_args* a, void* stream);
int aecf_pool_backward(const aecf_pool_desc* d, const aecf_pool_bwd_args* a, void* stream);

/* Stand-alone CurriculumMasking.forward on rows of length L (ref aecf/AECFLayer.py:130-283).
 * weights [rows,L] float32, any L.  mode 1 = train (uniforms required), 2 = eval.
 * Any output pointer may be NULL -- except mask_bits in train mode when L > 64 (rows that long keep their bits there). */
int aecf_curriculum_mask_forward(int64_t rows, int32_t L, int32_t mode, int32_t min_active,
                                 float base_mask_prob, float entropy_target, float eps,
                                 const float* weights, const float* uniforms,
                                 float* masked, float* entropy, float* mask_rate,
                                 uint8_t* mask_bits, void* stream);

/* Gradient of the stand-alone module's masked output and (eval mode) entropy w.r.t. weights.
 * d_masked [rows,L] or NULL, d_entropy [rows] or NULL, mask_bits from the forward (train). */
int aecf_curriculum_mask_backward(int64_t rows, int32_t L, int32_t mode, float eps,
                                  const float* weights, const uint8_t* mask_bits,
                                  const float* d_masked, const float* d_entropy,
                                  float* d_weights, void* stream);

/* CurriculumMasking.entropy_loss forward + backward (ref aecf/AECFLayer.py:285-314):
 * loss[0] = mean((nan_to_num(H) - log(last_seq_len)*entropy_target)^2); d_entropy = dloss/dH * upstream.
 * partial: scratch of aecf_entropy_loss_workspace_bytes(n). */
size_t aecf_entropy_loss_workspace_bytes(int64_t n);
/* loss[0] (dtype) = max(sum of the (n + 255) / 256 partial sums aecf_pool_forward left in ent_loss_partial, 0) / n */
int aecf_entropy_loss_from_partials(int64_t n, int32_t dtype, const float* partial, void* loss, void* stream);
/* entropy [n] and loss [1] have element type dtype (the reference computes the loss in the dtype of
 * info['entropy']); the arithmetic and d_entropy [n] are float32. */
int aecf_entropy_loss_fwd_bwd(int64_t n, int32_t dtype, int32_t last_seq_len, float entropy_target,
                              const void* entropy, float upstream, void* loss,
                              float* d_entropy, void* workspace, void* stream);

/* Projection-free single-head attention softmax(Q K^T * scale) V (ref aecf/AECFLayer.py:556-581).
 * q [B,S,E], k,v [B,T,E] dtype; out [B,S,E] dtype; probs [B,S,T] float32 saved for backward. */
int aecf_sdpa_forward(int64_t B, int32_t S, int32_t T, int32_t E, int32_t dtype, float scale,
                      const void* q, const void* k, const void* v, void* out, float* probs,
                      void* stream);
int aecf_sdpa_backward(int64_t B, int32_t S, int32_t T, int32_t E, int32_t dtype, float scale,
                       const void* q, const void* k, const void* v, const float* probs,
                       const void* dout, void* dq, void* dk, void* dv, void* stream);

/* ---- general multi-head attention ----
 * Everything nn.MultiheadAttention does for MultimodalAttentionPool.forward OUTSIDE the shared-query hot path
 * (ref aecf/AECFLayer.py:409-418, 480-521; torch functional.py:5836-5852, 6504-6519, 6554-6612): per-sample
 * queries, tgt_len > 1, key != value, attn_mask, key_padding_mask, attention dropout.  Batch-major tensors.
 *   Q = query W_q^T + b_q, K = key W_k^T + b_k, V = value W_v^T + b_v            (NT GEMMs)
 *   per (b,h): p = softmax(Q_h K_h^T / sqrt(hd) + attn_mask, key_padding_mask -> -inf); p' = dropout(p)
 *   o_h = p' V_h ;  y = o W_o^T + b_o ;  attn_w = mean_h p'   (the weights are returned AFTER dropout, as torch does)
 * Dropout takes explicit uniforms: keep = (u >= dropout_p), p' = p * keep / (1 - dropout_p)  (torch draws its own
 * Philox stream inside F.dropout, so the pattern is distribution-, not bit-, compatible).  tgt_len, src_len <= 4096 (the
 * score rows of a chunk of query positions live in LDS; longer targets are walked in chunks). */
typedef struct aecf_mha_desc {
    int64_t batch;
    int32_t tgt_len;      /* T */
    int32_t src_len;      /* S */
    int32_t embed_dim;    /* E <= 1024, multiple of 32 (f32) / 64 (bf16) */
    int32_t num_heads;    /* H, any divisor of E */
    int32_t dtype;        /* aecf_dtype of activations and weights */
    float dropout_p;      /* 0 = no dropout (then dropout_uniforms may be NULL) */
} aecf_mha_desc;

typedef struct aecf_mha_fwd_args {
    const void* query;           /* [B,T,E] dtype */
    const void* key;             /* [B,S,E] dtype */
    const void* value;           /* [B,S,E] dtype */
    const void* w_in;            /* [3E,E] */
    const void* b_in;            /* [3E] or NULL */
    const void* w_out;           /* [E,E] */
    const void* b_out;           /* [E] or NULL */
    const float* attn_mask;      /* additive float32 mask (-inf = blocked) [T,S] or [B*H,T,S], or NULL */
    int64_t attn_mask_stride;    /* 0 for a shared [T,S] mask, T*S for one per (b*H + h) */
    const uint8_t* key_padding_mask; /* [B,S] nonzero = ignore, or NULL */
    const float* dropout_uniforms;   /* [B*H,T,S] float32 U[0,1), or NULL when dropout_p == 0 */
    void* y;                     /* [B,T,E] dtype */
    float* attn_w;               /* [B,T,S] head-averaged weights (after dropout) */
    void* saved_q;               /* [B*T,E] dtype projections, kept for the backward */
    void* saved_k;               /* [B*S,E] */
    void* saved_v;               /* [B*S,E] */
    void* saved_o;               /* [B*T,E] pre-out-projection heads */
    float* saved_probs;          /* [B,H,T,S] softmax (before dropout) */
} aecf_mha_fwd_args;

typedef struct aecf_mha_bwd_args {
    const void* query;
    const void* key;
    const void* value;
    const void* w_in;
    const void* w_out;
    const float* dropout_uniforms;
    const void* dy;              /* [B,T,E] dtype */
    const float* d_attn_w;       /* [B,T,S] or NULL */
    const void* saved_q;
    const void* saved_k;
    const void* saved_v;
    const void* saved_o;
    const float* saved_probs;
    void* dquery;                /* [B,T,E] dtype */
    void* dkey;                  /* [B,S,E] dtype */
    void* dvalue;                /* [B,S,E] dtype */
    float* dw_in;                /* [3E,E] float32 */
    float* db_in;                /* [3E]   */
    float* dw_out;               /* [E,E]  */
    float* db_out;               /* [E]    */
    void* workspace;
    size_t workspace_bytes;
} aecf_mha_bwd_args;

int aecf_mha_check(const aecf_mha_desc* d);
size_t aecf_mha_bwd_workspace_bytes(const aecf_mha_desc* d);
int aecf_mha_forward(const aecf_mha_desc* d, const aecf_mha_fwd_args* a, void* stream);
int aecf_mha_backward(const aecf_mha_desc* d, const aecf_mha_bwd_args* a, void* stream);

/* ---- missing-modality front-end (ref xrays/train_xrays_example.py:156-177, 202-203) ----
 * One pass over one modality's feature rows feat [rows,dim]: rows with drop[r] != 0 are zeroed (the reference's
 * clone + masked write, :173-176) and present[r] = (||row||_2 > 1e-6) of the row AS WRITTEN (the reference's
 * torch.norm(...) > 1e-6 presence test, :202-203).  drop may be NULL (evaluation: presence only); out may equal
 * feat (in place) and may be NULL when drop is NULL (nothing to write).  Norm accumulated in float32. */
int aecf_modality_frontend(int64_t rows, int32_t dim, int32_t dtype, const void* feat, const uint8_t* drop,
                           void* out, uint8_t* present, void* stream);

/* ---- presence routing around the pool (ref xrays/train_xrays_example.py:205-234) ----
 * The reference routes rows with boolean masks + torch.where + torch.stack + index_put.  Here: one routing table built on
 * the device, then row moves driven by it.
 *   aecf_route_build: class of every row from the two presence vectors -- 0 both present (:205), 1 only a (:206),
 *     2 only b (:207), 3 neither -- its slot inside its class (ascending row order, what torch.where yields), the inverse
 *     lists index[c*rows + slot] = row for c = 0..2, and counts[4] (the only values the host reads back).
 *   aecf_rows_gather: up to 3 jobs dst_j[i] = src_j[index_j[i]], i < n_j, in ONE launch (rows of row_bytes bytes; pitches in
 *     bytes).  E.g. the [n_both, 2, E] pool input of :213-216 = two jobs writing the halves of one destination row.
 *   aecf_rows_select: dst[r] = src_{route[r]}[slot[r]] (class 3 or a NULL source: zeros); EVERY row of dst is written once,
 *     so no memset / index_put is needed (the fused [B, 2E] rows of :209-234; the backward of a gather). */
int aecf_route_build(int64_t rows, const uint8_t* present_a, const uint8_t* present_b, int32_t* route,
                     int32_t* slot, int32_t* index, int32_t* counts, void* stream);
int aecf_rows_gather(int32_t njobs, const void* const* src, const int64_t* src_pitch,
                     const int32_t* const* index, const int64_t* n, void* const* dst,
                     const int64_t* dst_pitch, int64_t row_bytes, void* stream);
int aecf_rows_select(int64_t rows, int64_t row_bytes, const int32_t* route, const int32_t* slot,
                     const void* const* src, const int64_t* src_pitch, void* dst, int64_t dst_pitch,
                     void* stream);

/* ---- static routing (every branch keeps all rows, so a whole step has fixed shapes and can be one HIP graph) ----
 *   aecf_front_pair: both modality front-ends, the missing-modality decisions (ref :156-172) and the row classes in one
 *     launch.  Decisions come from uniforms [3, rows] (a dropped if u0 < missing_prob, b if u1 < missing_prob; a row that
 *     would lose both keeps a when u2 > 0.5, else b), or from drop_a / drop_b (either may be NULL), or none (all three
 *     NULL).  present_x = not dropped and ||row|| > 1e-6 (float32; false for NaN rows, as torch.norm(...) > 1e-6 is, :202-203);
 *     out_x = the row if present_x else zeros (out_x != feat_x); cls as aecf_route_build's route.
 *   aecf_rows_select with slot == NULL reads src_c[r] (identity slots): fused[r] = the row of the branch that owns r.
 *   aecf_rows_split: its backward, dst_c[r] = src[r] if route[r] == c else zeros, c = 0..2 (NULL dst_c skipped), one launch. */
int aecf_front_pair(int64_t rows, int32_t dim_a, int32_t dim_b, int32_t dtype, const void* feat_a, const void* feat_b,
                    const float* uniforms, float missing_prob, const uint8_t* drop_a, const uint8_t* drop_b,
                    void* out_a, void* out_b, uint8_t* present_a, uint8_t* present_b, int32_t* cls, void* stream);
int aecf_rows_split(int64_t rows, int64_t row_bytes, const int32_t* route, const void* src, void* const* dst,
                    void* stream);

/* float32 -> bf16 (round to nearest even) of up to 8 tensors in one launch: the activation-dtype copies of float32 MASTER
 * parameters a mixed-precision step hands to aecf_pool_forward (w_in, b_in, w_out, b_out, query; ABI v9).  What
 * `p.to(torch.bfloat16)` does per tensor (ref: the reference trains in one dtype; torch's autocast makes these copies), as one
 * launch of 2048-element blocks.  src[i] / dst[i]: device pointers, numel[i] elements each; host arrays of length n <= 8. */
int aecf_cast_f32_to_bf16(int32_t n, const float* const* src, void* const* dst, const int64_t* numel, void* stream);

/* ---- the example trainer's optimiser step (ref xrays/train_xrays_example.py:322-323, 376: torch.optim.AdamW) ----
 * AdamW (decoupled weight decay, no amsgrad) over n float32 tensors in one launch per 24 tensors: arrays of n device
 * pointers (param, grad, exp_avg, exp_avg_sq: numel[i] floats each; step[i]: ONE float, the number of steps taken so far,
 * advanced on the device so that a captured step replays with a fresh count) and ticket = AECF_ADAMW_TICKET_WORDS * ((n + 23) / 24)
 * zero-initialised uint32 the launches use to find their last block (they leave them zero).  Arithmetic in float32; the bias corrections
 * 1 - beta^t are formed as -expm1(t ln beta) in float32 (relative error ~1e-7; torch uses double on the host or, capturable,
 * a float32 pow on the device).  beta = 0 is not supported (ln). */
#define AECF_ADAMW_TICKET_WORDS 65
int aecf_adamw_step(int32_t n, void* const* param, const void* const* grad, void* const* exp_avg,
                    void* const* exp_avg_sq, void* const* step, const int64_t* numel, void* ticket, float lr,
                    float beta1, float beta2, float eps, float weight_decay, void* stream);

/* ---- contrastive term (BASELINE.json north_star; NOT in the reference: SURVEY.md 8a row A9, build-defined) ----
 * Row-wise L2 normalisation zn = z / max(||z||, eps) and its backward dz = (dzn - zn (dzn.zn)) * inv_norm. */
int aecf_l2norm_forward(int64_t n, int32_t d, int32_t dtype, float eps, const void* z, void* zn,
                        float* inv_norm, void* stream);
int aecf_l2norm_backward(int64_t n, int32_t d, int32_t dtype, const void* zn, const float* inv_norm,
                         const float* dzn, void* dz, void* stream);
/* One InfoNCE direction, forward + backward in one call: local unit-norm queries q [rows,d] against all
 * (all-gathered) unit-norm keys k [cols,d]; the positive of local row i is key row_offset + i.
 *   loss_rows[i] = logsumexp_j(q_i.k_j / T) - q_i.k_pos / T                       float32 [rows]
 *   dq = coef/T * (softmax - onehot) k          float32 [rows,d]
 *   dk = coef/T * (softmax - onehot)^T q        float32 [cols,d]   (sum over ranks is the caller's reduce-scatter)
 * Two implementations, chosen by the workspace the caller hands over (the caller owns the memory):
 *   * tile-GEMM form (bf16, d % 64 == 0, temperature >= 0.025; workspace aecf_nce_workspace_bytes = rows x cols bf16 + O(rows
 *     + cols) floats): E = exp((q.k - 1)/T) is written once as bf16, its row sums come out of the same GEMM epilogue, the
 *     softmax weights are formed in place and dq / dk are two more tile GEMMs (transposed LDS reads: no transposed copies).
 *     6 rows cols d flops.  Rows of q and k MUST have L2 norm <= 1 (+ bf16 rounding): 1/T is used as the shift of every exponent.
 *   * streaming ("flash") form (bf16, d in {128, 256, 384, 512, 768, 1024}; workspace aecf_nce_stream_workspace_bytes =
 *     O(rows d)): the [rows, cols] logits are never materialised -- key tiles stream through LDS under an online max / sum per
 *     row, a second streaming pass forms dk from the saved log-sum-exp.  8 rows cols d flops; any norms, any temperature.
 *   A workspace of at least aecf_nce_workspace_bytes selects the first, a smaller one of at least
 *   aecf_nce_stream_workspace_bytes the second.  Otherwise (float32, other d): materialising float32 form, d % 64 == 0 and
 *   cols % 64 == 0. */
size_t aecf_nce_workspace_bytes(int64_t rows, int64_t cols, int32_t d, int32_t dtype);
size_t aecf_nce_stream_workspace_bytes(int64_t rows, int64_t cols, int32_t d, int32_t dtype);   /* 0: no streaming form */
int aecf_nce_fwd_bwd(int64_t rows, int64_t cols, int64_t row_offset, int32_t d, int32_t dtype,
                     float temperature, float coef, const void* q, const void* k, float* loss_rows,
                     float* dq, float* dk, void* workspace, size_t workspace_bytes, void* stream);

/* The loss side of the objective as ONE call (BASELINE.json north_star "contrastive + entropy_loss terms and their backward
 * are a second fused kernel"; README.md:205-208 of the reference for the entropy term): one InfoNCE direction exactly as
 * aecf_nce_fwd_bwd (bf16, streaming form) and, riding in its row-combine launch, CurriculumMasking.entropy_loss forward +
 * backward (ref aecf/AECFLayer.py:285-314) on entropy [n_entropy] float32:
 *   entropy_loss[0] = mean((nan_to_num(H) - log(last_seq_len) * entropy_target)^2), d_entropy = d loss / dH * entropy_upstream.
 * n_entropy == 0: contrastive term only.  Workspace: aecf_nce_workspace_bytes(rows, cols, d, AECF_BF16) (tile-GEMM form) or
 * aecf_nce_stream_workspace_bytes (streaming form), as for aecf_nce_fwd_bwd. */
int aecf_loss_fwd_bwd(int64_t rows, int64_t cols, int64_t row_offset, int32_t d, float temperature, float coef,
                      const void* q, const void* k, float* loss_rows, float* dq, float* dk, int64_t n_entropy,
                      int32_t last_seq_len, float entropy_target, const float* entropy, float entropy_upstream,
                      float* entropy_loss, float* d_entropy, void* workspace, size_t workspace_bytes, void* stream);


/* BOTH directions of the symmetric InfoNCE from ONE block of logits (build-defined, as above): local rows a [rows,d] (global
 * indices row_offset .. row_offset + rows) against all gathered keys b [cols,d], unit-norm bf16 rows, d % 64 == 0, T >= 0.025:
 *   L = coef * sum_i [ lse_j(a_i.b_j/T) - a_i.b_pos/T ]  +  coef * sum_i [ lse_i'(a_i'.b_pos/T) - a_i.b_pos/T ],  pos = row_offset + i
 * The second term's softmax runs down the COLUMNS of the global logits; a rank holds only its row block, so the column sums of
 * E = exp((a.b - 1)/T) are the one quantity ranks exchange:
 *   pass1: E (bf16, workspace), its row sums (workspace), and THIS rank's column sums col_sums [cols] (float32)
 *   caller: all-reduce (sum) of col_sums over the ranks (nothing to do on one rank)
 *   loss:  loss_rows[i] = both terms of local row i (float32 [rows]) from the summed column sums; optionally
 *          CurriculumMasking.entropy_loss forward + backward riding in the same launch (arguments as aecf_loss_fwd_bwd;
 *          n_entropy == 0: off).  Also prepares the normalisers the gradients need (kept in the workspace).
 *   grads: da = dL/da [rows,d]; db = this rank's share of dL/db [cols,d] (the sum over ranks is the caller's reduce-scatter), both
 *          multiplied by upstream[0] when `upstream` (a DEVICE float32 scalar: the gradient arriving at this term; no host read)
 *          is given, in float32 or -- one rounding of the float32 sums -- bf16 (grad_dtype).  May run long after `loss` (an
 *          autograd backward): the workspace must be left alone in between; it overwrites E with the softmax weights, so it
 *          runs once per pass1.
 * 6 rows cols d MFMA flops for both directions (2 for the logits, 2 + 2 for the gradient products through transposed LDS
 * reads) against 16 rows cols d for two calls of the streaming form.  Workspace: rows x cols bf16 + O(rows + cols) floats. */
size_t aecf_nce_sym_workspace_bytes(int64_t rows, int64_t cols, int32_t d);
int aecf_nce_sym_pass1(int64_t rows, int64_t cols, int32_t d, float temperature, const void* a, const void* b,
                       void* workspace, size_t workspace_bytes, float* col_sums, void* stream);
int aecf_nce_sym_loss(int64_t rows, int64_t cols, int64_t row_offset, int32_t d, float temperature, const void* a,
                      const void* b, const float* col_sums, void* workspace, size_t workspace_bytes, float* loss_rows,
                      int64_t n_entropy, int32_t last_seq_len, float entropy_target, const float* entropy,
                      float entropy_upstream, float* entropy_loss, float* d_entropy, void* stream);
int aecf_nce_sym_grads(int64_t rows, int64_t cols, int64_t row_offset, int32_t d, float temperature, float coef,
                       const void* a, const void* b, void* workspace, size_t workspace_bytes, const float* upstream,
                       int32_t grad_dtype, void* da, void* db, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AECF_HIP_H */
