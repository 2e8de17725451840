// extern "C" entry points of libaecf_hip.so (declared in include/aecf_hip.h).
// Host-side orchestration only: validates the description, carves the caller's workspace and
// enqueues the kernels on the caller's stream -- plain launches, so a caller that captures its stream (a whole training step
// as one HIP graph) gets them as nodes of ITS graph.  No device allocation, no synchronisation, no exceptions, no state.
#include <math.h>
#include <stdio.h>
#include <string>
#include <stdint.h>
#include <stdlib.h>


#include "../../include/aecf_hip.h"
#include "aecf_kernels.h"

using namespace aecf;

namespace {

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
int env_dx_reserve();          // tokens of the one debug knob AECF_DEBUG, read once per process (defined below;
bool env_no_gate_fusion();     //  env_no_ws / env_no_wide_tn / env_no_slab are declared in aecf_kernels.h: other files ask too)
bool env_fused_fwd();
inline int esize(int dtype) { return dtype == AECF_BF16 ? 2 : 4; }

struct FwdWs {
    size_t qs, a_f32, a_hi, a_lo, obuf, wv_frag, wo_frag, total;
};
FwdWs fwd_layout(const aecf_pool_desc* d) {
    FwdWs w;
    const size_t E = d->embed_dim, es = esize(d->dtype);
    size_t off = 0;
    w.qs = off;    off = align_up(off + E * 4);
    w.a_f32 = off; off = align_up(off + HPAD * E * 4);
    w.a_hi = off;  off = align_up(off + HPAD * E * es);
    w.a_lo = off;  off = align_up(off + HPAD * E * es);
    w.obuf = off;  off = align_up(off + (size_t)d->batch * E * es);
    w.wv_frag = off; off = align_up(off + E * E * es);          // fragment-major W_v, W_o (bf16, weight-stationary kernels)
    w.wo_frag = off; off = align_up(off + E * E * es);
    w.total = off;
    return w;
}

// what the backward derives from the parameters alone; lives at the head of the backward workspace or, when the caller
// passes saved_prep, in that buffer (filled by the forward's preparation launch)
struct PrepWs {
    size_t qs, a_f32, a_hi, a_lo, wvt, wot, wvt_frag, wot_frag, wv_frag, wo_frag, total;
};
PrepWs prep_layout(const aecf_pool_desc* d) {
    PrepWs w;
    const size_t E = d->embed_dim, es = esize(d->dtype);
    size_t off = 0;
    w.qs = off;     off = align_up(off + E * 4);
    w.a_f32 = off;  off = align_up(off + HPAD * E * 4);
    w.a_hi = off;   off = align_up(off + HPAD * E * es);
    w.a_lo = off;   off = align_up(off + HPAD * E * es);
    w.wvt = off;    off = align_up(off + E * E * es);
    w.wot = off;    off = align_up(off + E * E * es);
    w.wvt_frag = off; off = align_up(off + E * E * es);
    w.wot_frag = off; off = align_up(off + E * E * es);
    w.wv_frag = off; off = align_up(off + E * E * es);       // the FORWARD's fragment copies (ABI v9: with saved_prep they live here too,
    w.wo_frag = off; off = align_up(off + E * E * es);       //  so that a forward can reuse a whole preparation: AECF_PREP_READY)
    w.total = off;
    return w;
}

struct BwdWs {
    PrepWs prep;
    size_t dobuf, dsbuf, slab_o, slab_v, cs_o, cs_v, u_slab, u, dqp, dq_part, do_lo, total;
    int splits, u_splits, u_splits_cap;
    int64_t rows_per_split, u_rows_per_split;
};
BwdWs bwd_layout(const aecf_pool_desc* d, bool hilo = false) {
    BwdWs w;
    const size_t E = d->embed_dim, es = esize(d->dtype);
    const size_t B = (size_t)d->batch;
    const int tiles = (int)(((E + 127) / 128) * ((E + 127) / 128));
    int S = 512 / tiles;                                // one round of blocks: 2 per CU of the 128-row kernels (512 slots), 1 per
                                                        // CU of the 256-row pooled kernel (half the tiles, 256 slots); not more --
                                                        // at d = 768 ceil() made it 15 x 36 = 540 blocks, a second round for 28
    const int64_t max_s = (int64_t)((B + 255) / 256);   // but at least 256 batch rows per split
    if (S > max_s) S = (int)max_s;
    if (S < 1) S = 1;
    int64_t rps = (int64_t)((B + S - 1) / S);
    rps = (rps + 31) / 32 * 32;
    S = (int)((B + rps - 1) / rps);
    w.splits = S;
    w.rows_per_split = rps;
    {   // u = ds^T x: [16, E] per batch split, ~512 blocks of 8 waves (2 per CU)
        int Su = 512;
        const int64_t max_su = (int64_t)((B + 63) / 64);
        if (Su > max_su) Su = (int)max_su;
        if (Su < 1) Su = 1;
        int64_t urps = (int64_t)((B + Su - 1) / Su);
        urps = (urps + 15) / 16 * 16;
        w.u_splits = (int)((B + urps - 1) / urps);
        w.u_rows_per_split = urps;
    }
    w.prep = prep_layout(d);
    size_t off = w.prep.total;
    w.dobuf = off;  off = align_up(off + B * E * es);
    w.dsbuf = off;  off = align_up(off + B * d->num_heads * d->modalities * 4);
    // (AECF_HILO_GRADS, round 5: ONE launch per product on hi + lo tiles -- the same single slab sets as the default path)
    w.slab_o = off; off = align_up(off + (size_t)S * E * E * 4);
    w.slab_v = off; off = align_up(off + (size_t)S * E * E * 4);
    w.cs_o = off;   off = align_up(off + (size_t)S * E * 4);
    w.cs_v = off;   off = align_up(off + (size_t)S * E * 4);
    w.u_splits_cap = w.u_splits > 256 ? w.u_splits : 256;            // the head-split kernel writes up to 256 slabs
    w.u_slab = off; off = align_up(off + (size_t)w.u_splits_cap * HPAD * E * 4);
    w.u = off;      off = align_up(off + HPAD * E * 4);
    w.dqp = off;    off = align_up(off + E * 4);
    w.dq_part = off; off = align_up(off + (E / 16) * E * 4);
    w.do_lo = off;  if (hilo) off = align_up(off + B * E * es);
    w.total = off;
    return w;
}

// ---- AECF_PRECISE: float32-store form of the bf16 path (include/aecf_hip.h) ------------------------------------------
// forward workspace: [ordinary forward workspace of the bf16 description][w_out32 E*E][b_out32 E][o32 B*E (when the caller
// keeps no saved_o)]
struct PreciseFwdWs { size_t base, w_out32, b_out32, o32, total; };
PreciseFwdWs precise_fwd_layout(const aecf_pool_desc* d) {
    PreciseFwdWs w;
    const size_t E = d->embed_dim, B = (size_t)d->batch;
    size_t off = align_up(fwd_layout(d).total);
    w.base = 0;
    w.w_out32 = off; off = align_up(off + E * E * 4);
    w.b_out32 = off; off = align_up(off + E * 4);
    w.o32 = off;     off = align_up(off + B * E * 4);
    w.total = off;
    return w;
}
// backward workspace: [bf16 prep][x32][dy32][q32][w_in32][b_in32][w_out32][do32][float32 backward workspace]
struct PreciseBwdWs { size_t prep16, x32, dy32, q32, w_in32, b_in32, w_out32, do32, ws32, total; };
PreciseBwdWs precise_bwd_layout(const aecf_pool_desc* d) {
    PreciseBwdWs w;
    const size_t E = d->embed_dim, B = (size_t)d->batch, M = d->modalities;
    aecf_pool_desc d32 = *d;
    d32.dtype = AECF_F32;
    size_t off = 0;
    w.prep16 = off;  off = align_up(off + prep_layout(d).total);
    w.x32 = off;     off = align_up(off + B * M * E * 4);
    w.dy32 = off;    off = align_up(off + B * E * 4);
    w.q32 = off;     off = align_up(off + E * 4);
    w.w_in32 = off;  off = align_up(off + 3 * E * E * 4);
    w.b_in32 = off;  off = align_up(off + 3 * E * 4);
    w.w_out32 = off; off = align_up(off + E * E * 4);
    w.do32 = off;    off = align_up(off + B * E * 4);
    w.ws32 = off;    off = align_up(off + bwd_layout(&d32).total);
    w.total = off;
    return w;
}

inline void mark(void** ev, int i, hipStream_t s) {
    if (ev) (void)hipEventRecord((hipEvent_t)ev[i], s);
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? AECF_OK : AECF_ERR_LAUNCH; }

MaskCfg make_mask_cfg(int mode, int min_active, float p, float tau, float eps, int L) {
    MaskCfg c;
    c.mode = mode;
    c.min_active = min_active;
    c.base_mask_prob = p;
    c.entropy_target = tau;
    c.eps = eps;
    c.log_L = (float)log((double)L);
    c.inv_L = (float)(1.0 / (double)L);
    return c;
}

}  // namespace

extern "C" {

int aecf_abi_version(void) { return AECF_ABI_VERSION; }

const char* aecf_status_string(int status) {
    switch (status) {
        case AECF_OK: return "ok";
        case AECF_ERR_BAD_DIMS: return "bad dimensions";
        case AECF_ERR_UNSUPPORTED:
            return "configuration not supported by the HIP path (need 1<=M<=8, 1<=H<=16, E%64==0, head_dim%32==0 for "
                   "bf16 / %16==0 for f32)";
        case AECF_ERR_NULL_POINTER: return "required pointer is null";
        case AECF_ERR_WORKSPACE: return "workspace too small";
        case AECF_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}

const char* aecf_pool_stage_name(int backward, int stage) {
    static const char* fwd[AECF_FWD_STAGES] = {"prep", "gate", "vproj", "outproj"};
    static const char* bwd[AECF_BWD_STAGES] = {"prep", "dout", "dw_out", "dscore", "u", "dx", "dw_v", "finalize"};
    if (stage < 0) return nullptr;
    if (!backward) return stage < AECF_FWD_STAGES ? fwd[stage] : nullptr;
    return stage < AECF_BWD_STAGES ? bwd[stage] : nullptr;
}

int aecf_pool_check(const aecf_pool_desc* d) {
    if (!d) return AECF_ERR_NULL_POINTER;
    if (d->batch <= 0 || d->modalities <= 0 || d->embed_dim <= 0 || d->num_heads <= 0) return AECF_ERR_BAD_DIMS;
    if (d->embed_dim % d->num_heads != 0) return AECF_ERR_BAD_DIMS;
    if (d->dtype != AECF_BF16 && d->dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (d->modalities > 8 || d->num_heads > HPAD) return AECF_ERR_UNSUPPORTED;
    if (d->embed_dim % 64 != 0) return AECF_ERR_UNSUPPORTED;
    const int hd = d->embed_dim / d->num_heads;
    if (hd % (d->dtype == AECF_BF16 ? 32 : 16) != 0) return AECF_ERR_UNSUPPORTED;
    if (d->embed_dim > 1024) return AECF_ERR_UNSUPPORTED;
    if (d->mask_mode < 0 || d->mask_mode > 2) return AECF_ERR_BAD_DIMS;
    return AECF_OK;
}

size_t aecf_pool_fwd_workspace_bytes(const aecf_pool_desc* d) {
    if (aecf_pool_check(d) != AECF_OK) return 0;
    return fwd_layout(d).total;
}

size_t aecf_pool_bwd_workspace_bytes(const aecf_pool_desc* d) {
    if (aecf_pool_check(d) != AECF_OK) return 0;
    return bwd_layout(d).total;
}

size_t aecf_pool_precise_workspace_bytes(const aecf_pool_desc* d, int backward) {
    if (aecf_pool_check(d) != AECF_OK || d->dtype != AECF_BF16) return 0;
    return backward ? precise_bwd_layout(d).total : precise_fwd_layout(d).total;
}

int aecf_pool_wants_saved_v(const aecf_pool_desc* d) {
    if (aecf_pool_check(d) != AECF_OK) return 0;
    if (d->dtype == AECF_BF16 && !env_no_ws()) {
        BwdGArgs g;
        g.B = d->batch; g.M = d->modalities; g.E = d->embed_dim; g.H = d->num_heads; g.hd = d->embed_dim / d->num_heads;
        if (dsu_ws_chunks(g) > 0) return 0;        // the score gradient comes from x (dsu_ws_kernel): nothing to save
    }
    return 1;
}

// AECF_HILO_GRADS is built for the bf16 shapes whose value projection runs on the weight-stationary kernel in its
// per-sample form (the kernels that can write the low part of o and of do)
static bool hilo_supported(const aecf_pool_desc* d) {
    if (aecf_pool_check(d) != AECF_OK || d->dtype != AECF_BF16 || env_no_ws()) return false;
    GemmNtArgs v;
    v.a = nullptr; v.w = nullptr; v.bias = nullptr; v.c = nullptr; v.probs = nullptr; v.R = d->batch; v.N = d->embed_dim;
    v.K = d->embed_dim; v.lda = (int64_t)d->modalities * d->embed_dim; v.M = d->modalities; v.H = d->num_heads;
    v.hd = d->embed_dim / d->num_heads; v.pooled = 1; v.out_f32 = 0; v.v_out = nullptr;
    if (!gemm_ws_supported(v)) return false;
    if (d->modalities == 4 && d->embed_dim > 512) return false;         // (the flat-row form has no per-sample o in a lane)
    GemmNtArgs y = v;
    y.pooled = 0; y.M = 1; y.lda = d->embed_dim;
    if (!gemm_ws_supported(y)) return false;
    BwdGArgs g;                                                         // ... and the score gradient on dsu_ws_kernel (do_hi + do_lo)
    g.B = d->batch; g.M = d->modalities; g.E = d->embed_dim; g.H = d->num_heads; g.hd = d->embed_dim / d->num_heads;
    if (dsu_ws_chunks(g) <= 0) return false;
    GemmTnArgs t;                                                       // ... and the two batch reductions on hi + lo tiles in one launch each
    t.lhs = t.rhs = nullptr; t.probs = nullptr; t.dsbuf = nullptr; t.out = nullptr; t.colsum = nullptr; t.u = nullptr;
    t.B = d->batch; t.M = d->modalities; t.E = d->embed_dim; t.H = d->num_heads; t.hd = d->embed_dim / d->num_heads; t.Ej = 0;
    const BwdWs L = bwd_layout(d, true);
    t.splits = L.splits; t.rows_per_split = L.rows_per_split; t.u_splits = 0; t.u_rows_per_split = 0;
    t.pooled = 1;
    if (!gemm_tn_hilo_supported(t)) return false;
    t.pooled = 0; t.M = 1;
    return gemm_tn_hilo_supported(t);
}

size_t aecf_pool_hilo_bwd_workspace_bytes(const aecf_pool_desc* d) {
    if (!hilo_supported(d)) return 0;
    return bwd_layout(d, true).total;
}

size_t aecf_pool_prep_bytes(const aecf_pool_desc* d) {
    if (aecf_pool_check(d) != AECF_OK) return 0;
    return prep_layout(d).total;
}

}  // extern "C"

namespace {

// validation + every launch of the forward on stream s (the caller's stream)
int pool_forward_on(const aecf_pool_desc* d, const aecf_pool_fwd_args* a, hipStream_t s) {
    int st = aecf_pool_check(d);
    if (st != AECF_OK) return st;
    const bool precise = a && (a->flags & AECF_PRECISE);
    if (precise && d->dtype != AECF_BF16) return AECF_ERR_UNSUPPORTED;
    if (precise && a->workspace_bytes < precise_fwd_layout(d).total) return AECF_ERR_WORKSPACE;
    if (!a || !a->x || !a->query || !a->w_in || !a->w_out || !a->y || !a->attn_w || !a->saved_probs || !a->workspace)
        return AECF_ERR_NULL_POINTER;
    const bool prep_ready = (a->flags & AECF_PREP_READY) != 0;
    if (prep_ready && !a->saved_prep) return AECF_ERR_NULL_POINTER;
    const bool draw = d->mask_mode == 1 && !a->uniforms && (a->flags & AECF_DRAW_UNIFORMS);
    if (d->mask_mode == 1 && !a->uniforms && !draw) return AECF_ERR_NULL_POINTER;
    if (draw && (a->philox_threads == 0 || a->philox_threads % 256 != 0 || a->philox_offset % 4 != 0 || a->philox_element0 < 0))
        return AECF_ERR_BAD_DIMS;
    const FwdWs L = fwd_layout(d);
    if (a->workspace_bytes < L.total) return AECF_ERR_WORKSPACE;
    char* ws = (char*)a->workspace;
    const int E = d->embed_dim, H = d->num_heads, M = d->modalities, hd = E / H, es = esize(d->dtype);
    // with saved_prep the parameter-only products go there (the backward reads them back) together with the backward's
    // own transposes and fragment copies; the forward's fragment copies stay in its workspace
    const PrepWs P = prep_layout(d);
    char* pb = (char*)a->saved_prep;
    float* qs = pb ? (float*)(pb + P.qs) : (float*)(ws + L.qs);
    float* a_f32 = pb ? (float*)(pb + P.a_f32) : (float*)(ws + L.a_f32);
    void* a_hi = pb ? (void*)(pb + P.a_hi) : (void*)(ws + L.a_hi);
    void* a_lo = pb ? (void*)(pb + P.a_lo) : (void*)(ws + L.a_lo);
    const PreciseFwdWs PW = precise ? precise_fwd_layout(d) : PreciseFwdWs{};
    void* o = a->saved_o ? a->saved_o : (void*)(ws + (precise ? PW.o32 : L.obuf));     // (precise: float32 [B,E])
    const float scale = sqrtf(1.0f / (float)hd);     // torch functional.py:6577 q * sqrt(1/head_dim)

    void** ev = a->stage_events;
    mark(ev, 0, s);
    // fragment-major copies of W_v and W_o for the weight-stationary kernels (bf16, E in {256, 512, 768, 1024})
    const bool frag = d->dtype == AECF_BF16 && (E == 256 || E == 512 || E == 768 || E == 1024) && !env_no_ws();
    FragJobs fj;
    if (frag) {
        fj.n = 2;
        fj.src[0] = (const char*)a->w_in + (size_t)2 * E * E * es; fj.dst[0] = pb ? pb + P.wv_frag : ws + L.wv_frag; fj.transposed[0] = 0;
        fj.src[1] = a->w_out;                                      fj.dst[1] = pb ? pb + P.wo_frag : ws + L.wo_frag; fj.transposed[1] = 0;
        if (pb) {
            fj.n = 4;
            fj.src[2] = fj.src[0]; fj.dst[2] = pb + P.wvt_frag; fj.transposed[2] = 1;
            fj.src[3] = a->w_out;  fj.dst[3] = pb + P.wot_frag; fj.transposed[3] = 1;
        }
    }
    const char* w_v_src = (const char*)a->w_in + (size_t)2 * E * E * es;
    // (AECF_PREP_READY: saved_prep still holds the preparation of these very parameters -- an inference loop, gradient
    //  accumulation over micro-batches: nothing to launch)
    if (!prep_ready)
        launch_prep_all(d->dtype, a->w_in, a->b_in, a->query, scale, qs, a_f32, a_hi, a_lo, pb ? w_v_src : nullptr,
                        pb ? pb + P.wvt : nullptr, pb ? a->w_out : nullptr, pb ? pb + P.wot : nullptr, E, H, fj, s);
    mark(ev, 1, s);

    GateArgs g;
    g.x = a->x; g.a_hi = a_hi; g.a_lo = a_lo; g.kpm = a->key_padding_mask; g.uniforms = a->uniforms;
    g.probs = a->saved_probs; g.attn_w = a->attn_w; g.masked_w = a->masked_w; g.entropy = a->entropy;
    g.mask_rate = a->mask_rate; g.i_attn_w = a->info_attn_w; g.i_masked_w = a->info_masked_w;
    g.i_entropy = a->info_entropy; g.i_mask_rate = a->info_mask_rate;
    g.i_target = d->mask_mode == 1 ? a->info_target_entropy : nullptr; g.target_value = a->target_entropy_value;
    g.B = d->batch; g.M = M; g.E = E; g.H = H;
    g.mask = make_mask_cfg(d->mask_mode, d->min_active, d->base_mask_prob, d->entropy_target, d->eps, M);
    if (draw) {
        g.ph.seed = a->philox_seed; g.ph.offset = a->philox_offset; g.ph.threads = a->philox_threads;
        g.ph.elem0 = (unsigned long long)a->philox_element0;
    }
    GemmNtArgs v;
    v.a = a->x; v.w = (const char*)a->w_in + (size_t)2 * E * E * es;
    v.bias = a->b_in ? (const char*)a->b_in + (size_t)2 * E * es : nullptr;
    v.c = o; v.probs = a->saved_probs; v.R = d->batch; v.N = E; v.K = E; v.lda = (int64_t)M * E;
    v.M = M; v.H = H; v.hd = hd; v.pooled = 1; v.out_f32 = precise ? 1 : 0; v.v_out = precise ? nullptr : a->saved_v;
    if (frag) v.w_frag = pb ? pb + P.wv_frag : ws + L.wv_frag;
    const bool hilo = (a->flags & AECF_HILO_GRADS) != 0;
    if (hilo) {
        if (precise || !hilo_supported(d)) return AECF_ERR_UNSUPPORTED;
        if (!a->saved_o || !a->saved_o_lo) return AECF_ERR_NULL_POINTER;
        v.c_lo = a->saved_o_lo;
    }
    if (env_fused_fwd() && !precise && !hilo && row_fwd_supported(d->dtype, E, M, H)) {
        // ONE kernel from x to y: scores, softmax, statistics, value projection, pooling, out-projection (aecf_row_fwd.hip)
        GemmNtArgs yo;
        yo.a = o; yo.w = a->w_out; yo.bias = a->b_out; yo.c = a->y; yo.probs = nullptr; yo.R = d->batch; yo.N = E; yo.K = E;
        yo.lda = E; yo.M = 1; yo.H = H; yo.hd = hd; yo.pooled = 0; yo.out_f32 = 0; yo.v_out = nullptr;
        v.c = a->saved_o;                                  // kept only when the caller wants it (backward)
        mark(ev, 2, s);
        launch_row_fwd(g, v, yo, s);
        if (d->mask_mode == 1 && a->ent_loss_partial) {
            if (a->info_entropy) launch_entropy_partials(d->dtype, d->batch, a->target_entropy_value, a->info_entropy, a->ent_loss_partial, s);
            else if (a->entropy) launch_entropy_partials(AECF_F32, d->batch, a->target_entropy_value, a->entropy, a->ent_loss_partial, s);
            if (a->ent_loss) launch_entropy_from_partials(d->dtype, d->batch, a->ent_loss_partial, a->ent_loss, s);
        }
        mark(ev, 3, s);
        mark(ev, 4, s);
        return launch_status();
    }
    // bf16, shapes of the weight-stationary kernel, M <= 3: the scores are formed inside the value projection (one pass
    // over x for both) and the per-sample statistics follow from the saved weights; otherwise the gate kernel runs first
    bool fuse_gate = d->dtype == AECF_BF16 && M <= 3 && !env_no_gate_fusion() && !env_no_ws();   // (A/B timing switches)
    if (fuse_gate) {
        v.g_ahi = a_hi; v.g_alo = a_lo; v.g_kpm = a->key_padding_mask;
        if (!gemm_ws_supported(v)) { fuse_gate = false; v.g_ahi = v.g_alo = nullptr; v.g_kpm = nullptr; }
    }
    // entropy-regulariser partial sums (optional): by the statistics kernel where it runs, else by one small launch behind the
    // kernel that wrote the entropies (the caller may rely on them either way)
    float* ent_partial = (d->mask_mode == 1 && !precise) ? a->ent_loss_partial : nullptr;
    auto partials_after = [&]() {
        if (!ent_partial) return;
        if (a->info_entropy) launch_entropy_partials(d->dtype, d->batch, a->target_entropy_value, a->info_entropy, ent_partial, s);
        else if (a->entropy) launch_entropy_partials(AECF_F32, d->batch, a->target_entropy_value, a->entropy, ent_partial, s);
    };
    if (!fuse_gate) { launch_gate_fwd(d->dtype, g, s); partials_after(); }
    mark(ev, 2, s);
    launch_gemm_nt(d->dtype, v, s);
    if (fuse_gate) { g.ent_partial = ent_partial; launch_gate_stats(d->dtype, g, s); }
    mark(ev, 3, s);

    GemmNtArgs y;
    y.a = o; y.w = a->w_out; y.bias = a->b_out; y.c = a->y; y.probs = nullptr; y.R = d->batch; y.N = E; y.K = E;
    y.lda = E; y.M = 1; y.H = H; y.hd = hd; y.pooled = 0; y.out_f32 = 0; y.v_out = nullptr;
    if (precise) {
        // the out-projection consumes an intermediate (o): float32 operands, exact float32 MFMA, float32 y
        launch_cast_bf16_f32(a->w_out, (float*)(ws + PW.w_out32), (int64_t)E * E, s);
        launch_cast_bf16_f32(a->b_out, (float*)(ws + PW.b_out32), E, s);
        y.w = ws + PW.w_out32;
        y.bias = a->b_out ? (const void*)(ws + PW.b_out32) : nullptr;
        launch_gemm_nt(AECF_F32, y, s);
        mark(ev, 4, s);
        return launch_status();
    }
    if (frag) y.w_frag = pb ? pb + P.wo_frag : ws + L.wo_frag;
    // the entropy regulariser's final sum rides in the out-projection launch where the weight-stationary kernel runs it
    // (its first block adds the statistics kernel's partial sums: no launch of its own), else it is one small launch
    const bool want_loss = ent_partial && a->ent_loss;
    const bool loss_rides = want_loss && d->dtype == AECF_BF16 && !env_no_ws() && gemm_ws_supported(y);
    if (loss_rides) {
        y.ent_partial = ent_partial; y.ent_nblk = (int)((d->batch + 255) / 256); y.ent_inv_n = 1.0f / (float)d->batch;
        y.ent_loss = a->ent_loss;
    }
    launch_gemm_nt(d->dtype, y, s);
    if (want_loss && !loss_rides) launch_entropy_from_partials(d->dtype, d->batch, ent_partial, a->ent_loss, s);
    mark(ev, 4, s);
    return launch_status();
}

int pool_backward_on(const aecf_pool_desc* d, const aecf_pool_bwd_args* a, hipStream_t s, const void* do_ready = nullptr);

// AECF_PRECISE backward: do = dy W_o on the bf16 kernel (exact bf16 operands) stored in float32; everything that consumes
// an intermediate (do, o) runs on the float32 kernels from float32 copies of the bf16 inputs
int pool_backward_precise(const aecf_pool_desc* d, const aecf_pool_bwd_args* a, hipStream_t s) {
    if (d->dtype != AECF_BF16 || a->grad_dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!a->x || !a->query || !a->w_in || !a->w_out || !a->dy || !a->workspace) return AECF_ERR_NULL_POINTER;
    const PreciseBwdWs L = precise_bwd_layout(d);
    if (a->workspace_bytes < L.total) return AECF_ERR_WORKSPACE;
    char* ws = (char*)a->workspace;
    const int E = d->embed_dim, H = d->num_heads, M = d->modalities, hd = E / H;
    const int64_t B = d->batch;
    launch_cast_bf16_f32(a->x, (float*)(ws + L.x32), B * M * E, s);
    launch_cast_bf16_f32(a->dy, (float*)(ws + L.dy32), B * E, s);
    launch_cast_bf16_f32(a->query, (float*)(ws + L.q32), E, s);
    launch_cast_bf16_f32(a->w_in, (float*)(ws + L.w_in32), (int64_t)3 * E * E, s);
    launch_cast_bf16_f32(a->b_in, (float*)(ws + L.b_in32), 3 * E, s);
    launch_cast_bf16_f32(a->w_out, (float*)(ws + L.w_out32), (int64_t)E * E, s);
    // bf16 preparation (W_o^T and its fragment copy), then do = dy W_o with float32 stores
    const PrepWs P = prep_layout(d);
    char* pb = ws + L.prep16;
    const float scale = sqrtf(1.0f / (float)hd);
    const char* w_v = (const char*)a->w_in + (size_t)2 * E * E * 2;
    const bool frag = (E == 256 || E == 512 || E == 768 || E == 1024) && !env_no_ws();
    FragJobs fj;
    if (frag) {
        fj.n = 2;
        fj.src[0] = w_v;      fj.dst[0] = pb + P.wvt_frag; fj.transposed[0] = 1;
        fj.src[1] = a->w_out; fj.dst[1] = pb + P.wot_frag; fj.transposed[1] = 1;
    }
    launch_prep_all(AECF_BF16, a->w_in, a->b_in, a->query, scale, (float*)(pb + P.qs), (float*)(pb + P.a_f32), pb + P.a_hi,
                    pb + P.a_lo, w_v, pb + P.wvt, a->w_out, pb + P.wot, E, H, fj, s);
    GemmNtArgs g;
    g.a = a->dy; g.w = pb + P.wot; g.bias = nullptr; g.c = ws + L.do32; g.probs = nullptr; g.R = B; g.N = E; g.K = E; g.lda = E;
    g.M = 1; g.H = H; g.hd = hd; g.pooled = 0; g.out_f32 = 1; g.v_out = nullptr;
    if (frag) g.w_frag = pb + P.wot_frag;
    launch_gemm_nt(AECF_BF16, g, s);
    aecf_pool_desc d32 = *d;
    d32.dtype = AECF_F32;
    aecf_pool_bwd_args a32 = *a;
    a32.x = ws + L.x32; a32.query = ws + L.q32; a32.w_in = ws + L.w_in32; a32.b_in = a->b_in ? (const void*)(ws + L.b_in32) : nullptr;
    a32.w_out = ws + L.w_out32; a32.dy = ws + L.dy32; a32.saved_v = nullptr; a32.saved_prep = nullptr;
    a32.workspace = ws + L.ws32; a32.workspace_bytes = a->workspace_bytes - L.ws32; a32.flags = 0; a32.stage_events = nullptr;
    return pool_backward_on(&d32, &a32, s, ws + L.do32);
}

int pool_backward_on(const aecf_pool_desc* d, const aecf_pool_bwd_args* a, hipStream_t s, const void* do_ready) {
    int st = aecf_pool_check(d);
    if (st != AECF_OK) return st;
    if (a && (a->flags & AECF_PRECISE)) return pool_backward_precise(d, a, s);
    if (!a || !a->x || !a->query || !a->w_in || !a->w_out || !a->dy || !a->saved_probs || !a->saved_o || !a->dx ||
        !a->dquery || !a->dw_in || !a->db_in || !a->dw_out || !a->db_out || !a->workspace)
        return AECF_ERR_NULL_POINTER;
    if (a->d_entropy && !a->attn_w) return AECF_ERR_NULL_POINTER;
    if (a->grad_dtype != AECF_F32 && !(a->grad_dtype == AECF_BF16 && d->dtype == AECF_BF16)) return AECF_ERR_UNSUPPORTED;
    const bool hilo = (a->flags & AECF_HILO_GRADS) != 0;
    if (hilo && (!hilo_supported(d) || do_ready)) return AECF_ERR_UNSUPPORTED;
    if (hilo && !a->saved_o_lo) return AECF_ERR_NULL_POINTER;
    const BwdWs L = bwd_layout(d, hilo);
    if (a->workspace_bytes < L.total) return AECF_ERR_WORKSPACE;
    char* ws = (char*)a->workspace;
    const int E = d->embed_dim, H = d->num_heads, M = d->modalities, hd = E / H, es = esize(d->dtype);
    const int64_t B = d->batch;
    const PrepWs& P = L.prep;
    char* pb = a->saved_prep ? (char*)const_cast<void*>(a->saved_prep) : ws;
    float* qs = (float*)(pb + P.qs);
    float* a_f32 = (float*)(pb + P.a_f32);
    void* wvt = pb + P.wvt;
    void* wot = pb + P.wot;
    void* dobuf = do_ready ? const_cast<void*>(do_ready) : (void*)(ws + L.dobuf);
    float* dsbuf = (float*)(ws + L.dsbuf);
    float* u = (float*)(ws + L.u);
    const float scale = sqrtf(1.0f / (float)hd);
    const char* w_v = (const char*)a->w_in + (size_t)2 * E * E * es;

    void** ev = a->stage_events;
    mark(ev, 0, s);
    const bool frag = d->dtype == AECF_BF16 && (E == 256 || E == 512 || E == 768 || E == 1024) && !env_no_ws();
    FragJobs fj;
    if (!a->saved_prep) {                             // (with saved_prep the forward already produced all of this)
        if (frag) {                                   // fragment-major W_v^T (dx) and W_o^T (dout)
            fj.n = 2;
            fj.src[0] = w_v;      fj.dst[0] = pb + P.wvt_frag; fj.transposed[0] = 1;
            fj.src[1] = a->w_out; fj.dst[1] = pb + P.wot_frag; fj.transposed[1] = 1;
        }
        launch_prep_all(d->dtype, a->w_in, a->b_in, a->query, scale, qs, a_f32, pb + P.a_hi, pb + P.a_lo, w_v, wvt, a->w_out,
                        wot, E, H, fj, s);
    }
    mark(ev, 1, s);

    // do = dy W_o   (NT GEMM against W_o^T)
    GemmNtArgs g;
    g.a = a->dy; g.w = wot; g.bias = nullptr; g.c = dobuf; g.probs = nullptr; g.R = B; g.N = E; g.K = E; g.lda = E;
    g.M = 1; g.H = H; g.hd = hd; g.pooled = 0; g.out_f32 = 0; g.v_out = nullptr;
    if (frag) g.w_frag = pb + P.wot_frag;
    if (hilo) g.c_lo = ws + L.do_lo;
    if (!do_ready) launch_gemm_nt(d->dtype, g, s);
    mark(ev, 2, s);

    // dW_o = dy^T o, db_o = colsum(dy)
    GemmTnArgs t1;
    t1.lhs = a->dy; t1.rhs = a->saved_o; t1.probs = nullptr; t1.dsbuf = nullptr; t1.out = (float*)(ws + L.slab_o);
    t1.colsum = (float*)(ws + L.cs_o); t1.u = nullptr; t1.B = B; t1.M = 1; t1.E = E; t1.H = H; t1.hd = hd;
    t1.Ej = 0; t1.splits = L.splits; t1.rows_per_split = L.rows_per_split; t1.pooled = 0;
    t1.u_splits = 0; t1.u_rows_per_split = 0;
    // (launched here, between dout and the score gradient; moving it behind dW_v so that `do` is consumed while it may still
    //  sit in the 256 MB memory-side cache measured 0.5 % SLOWER, 4 of 4 same-box pairs: profiles/r05_c2_experiments.txt)
    if (hilo) {                                       // dy^T (o_hi + o_lo): both rhs tiles of a step in one launch
        t1.rhs_lo = a->saved_o_lo;
        launch_gemm_tn_hilo(t1, s);
    } else {
        launch_gemm_tn(d->dtype, t1, s);
    }
    const int gb = a->grad_dtype == AECF_BF16 ? 1 : 0;
    mark(ev, 3, s);

    BwdGArgs g2;
    g2.x = a->x; g2.dobuf = dobuf; g2.wvt = wvt; g2.probs = a->saved_probs; g2.d_attn_w = a->d_attn_w;
    g2.d_entropy = a->d_entropy; g2.attn_w = a->attn_w; g2.dsbuf = dsbuf; g2.a_f32 = a_f32; g2.dx = a->dx;
    g2.B = B; g2.M = M; g2.E = E; g2.H = H; g2.hd = hd; g2.log_M = (float)log((double)M);
    if (frag) g2.wvt_frag = pb + P.wvt_frag;
    // score gradient.  bf16 shapes of the head-split weight-stationary kernel: ds AND u = ds^T x in one pass over (do, x),
    // nothing saved by the forward; otherwise from the saved V (memory-bound dot), else by recomputing W_v^T do per head
    int dsu_chunks = 0;
    if (hilo) g2.do_lo = ws + L.do_lo;                // (read by dsu_ws_kernel only: other shapes keep the default key-side accuracy)
    if (d->dtype == AECF_BF16 && !env_no_ws() && dsu_ws_chunks(g2) > 0 && dsu_ws_chunks(g2) <= L.u_splits_cap)
        dsu_chunks = launch_dsu_ws(g2, (float*)(ws + L.u_slab), s);
    else if (!(a->saved_v && launch_dscore_v(d->dtype, g2, a->saved_v, s)))
        launch_bwd_g(d->dtype, g2, false, s);
    mark(ev, 4, s);

    // dW_v = do^T pooled, db_v = colsum(do), u = ds^T x
    GemmTnArgs t2;
    t2.lhs = dobuf; t2.rhs = a->x; t2.probs = a->saved_probs; t2.dsbuf = dsbuf; t2.out = (float*)(ws + L.slab_v);
    t2.colsum = (float*)(ws + L.cs_v); t2.u = (float*)(ws + L.u_slab); t2.B = B; t2.M = M; t2.E = E; t2.H = H;
    t2.hd = hd; t2.Ej = 0; t2.splits = L.splits; t2.rows_per_split = L.rows_per_split; t2.pooled = 1;
    t2.u_splits = L.u_splits; t2.u_rows_per_split = L.u_rows_per_split;
    // the key-side batch reduction u = ds^T x as its own pass over x where the score-gradient kernel did not form it: BEFORE
    // the dx kernel (round 4; it needs ds and x only), so that dx can add its slabs up like the fused kernel's
    if (!dsu_chunks) {
        t2.parts = 2;
        launch_gemm_tn(d->dtype, t2, s);
    }
    mark(ev, 5, s);
    // input gradient: here, or -- when the caller wants to be told the moment the parameter gradients are final -- last
    const bool dx_last = a->param_grads_event != nullptr;
    // ... and then with a few CUs left free: the dx kernel otherwise takes every CU's whole register file for its one
    // block, and the collective's workgroups could only start as those retire (AECF_DX_RESERVE_CUS, default 16 of 256;
    // UNMEASURED here -- no multi-GPU box -- it costs the dx kernel ~6 % and is meant to buy the all-reduce its overlap)
    if (dx_last) g2.cu_budget = 256 - env_dx_reserve();
    // where the weight-stationary dx kernel runs between the score gradient and the finalize launch, it also adds up the
    // u slabs (a side job of its weight prologue): the finalize launch then depends on no reduction launch
    bool u_reduced = false;
    if (!dx_last) { g2.u_slab_in = (const float*)(ws + L.u_slab); g2.u_out = u; g2.u_nslab = dsu_chunks ? dsu_chunks : L.u_splits; }
    auto run_dx = [&]() {
        if (d->dtype == AECF_BF16 && launch_dx_ws(g2, s)) { u_reduced = g2.u_slab_in != nullptr; return; }
        launch_bwd_g(d->dtype, g2, true, s);
    };
    if (!dx_last) run_dx();
    mark(ev, 6, s);

    t2.parts = 1;
    // dq' = scale W_k u (every remaining gradient of the tail hangs on it) rides in this launch where u is already reduced
    DqpJob dq;
    dq.w_k = (const char*)a->w_in + (size_t)E * E * es; dq.u = u; dq.dqp = (float*)(ws + L.dqp); dq.scale = scale; dq.E = E; dq.hd = hd;
    const bool dqp_rides = u_reduced && d->dtype == AECF_BF16;
    if (dqp_rides) t2.dq = dq;
    if (hilo) {
        // dW_v = do_hi^T pooled_hi + do_hi^T pooled_lo + do_lo^T pooled_hi, db_v = colsum(do_hi + do_lo): one launch that lands
        // both lhs tiles of a step and splits the pooled rows where it forms them (aecf_gemm_tn_hilo.hip)
        t2.lhs_lo = ws + L.do_lo;
        launch_gemm_tn_hilo(t2, s);
    } else {
        launch_gemm_tn(d->dtype, t2, s);
    }
    mark(ev, 7, s);

    ReduceSegs rs;
    for (int i = 0; i < ReduceSegs::N; ++i) rs.splits[i] = L.splits;
    rs.splits[4] = dsu_chunks ? dsu_chunks : L.u_splits;
    const size_t gsz = gb ? 2 : 4;                                    // bytes per parameter-gradient element
    for (int i = 0; i < ReduceSegs::N; ++i) rs.dst_bf16[i] = gb;
    rs.dst_bf16[4] = 0;                                               // u stays float32 (internal)
    const float gscale = a->grad_scale != 0.f ? a->grad_scale : 1.f;  // (ABI v9: 1 / world of a data-parallel caller)
    for (int i = 0; i < 4; ++i) rs.scale[i] = gscale;
    rs.src[0] = (const float*)(ws + L.slab_o); rs.dst[0] = a->dw_out;                              rs.n[0] = (int64_t)E * E;
    rs.src[1] = (const float*)(ws + L.cs_o);   rs.dst[1] = a->db_out;                              rs.n[1] = E;
    rs.src[2] = (const float*)(ws + L.slab_v); rs.dst[2] = (char*)a->dw_in + (size_t)2 * E * E * gsz; rs.n[2] = (int64_t)E * E;
    rs.src[3] = (const float*)(ws + L.cs_v);   rs.dst[3] = (char*)a->db_in + (size_t)2 * E * gsz;  rs.n[3] = E;
    rs.src[4] = (const float*)(ws + L.u_slab); rs.dst[4] = u;                                      rs.n[4] = (int64_t)H * E;
    if (!u_reduced) {                                                  // u first: every finalize block reads all of it
        if (!dqp_rides && rs.splits[4] <= 8) {                         // a few slabs: the dq' launch adds them up as it reads them
            dq.u_slab = (const float*)(ws + L.u_slab); dq.u_nslab = rs.splits[4]; dq.u_out = u;
        } else {
            ReduceSegs ru = rs;
            for (int i = 0; i < 4; ++i) ru.n[i] = 0;
            launch_reduce_segments(ru, s);
        }
    }
    rs.n[4] = 0;
    if (!dqp_rides) launch_dqp(d->dtype, dq, s);

    FinalizeArgs f;
    f.w_in = a->w_in; f.query = a->query; f.qs = qs; f.u = u; f.dqp = (float*)(ws + L.dqp);
    f.dq_part = (float*)(ws + L.dq_part); f.dw_in = a->dw_in;
    f.db_in = a->db_in; f.dquery = a->dquery; f.E = E; f.H = H; f.hd = hd; f.scale = scale; f.grad_bf16 = gb;
    f.gscale = gscale;
    launch_finalize_all(d->dtype, f, rs, s);
    if (dx_last) {
        (void)hipEventRecord((hipEvent_t)a->param_grads_event, s);
        run_dx();
    }
    mark(ev, 8, s);
    return launch_status();
}

// ONE debug knob, read once per process (tests and A/B timing; never on the call path): AECF_DEBUG = comma-separated tokens
//   no_ws               tiled round-1 kernels instead of the weight-stationary ones
//   no_gate_fusion      scores / softmax as their own kernel instead of inside the value projection
//   no_wide_tn          128-row tiles instead of the 1024-thread 256-row form of the pooled batch reduction
//   no_slab             the two-barrier gated value projection instead of its column-slab form (d = 512)
//   fused_fwd           the one-kernel forward north_star names (aecf_row_fwd.hip; measured slower: profiles/r03_c2_fusedfwd_*)
//   dx_reserve=N        CUs the dx kernel leaves free when it runs beside a collective (default 16, 0..128)
struct EnvSwitches {
    bool no_ws = false, no_gate_fusion = false, no_wide_tn = false, no_slab = false, fused_fwd = false;
    int dx_reserve = 16;
};
const EnvSwitches& env_switches() {
    static const EnvSwitches e = [] {
        EnvSwitches v;
        const char* env = getenv("AECF_DEBUG");
        if (!env) return v;
        std::string all(env);
        size_t pos = 0;
        while (pos <= all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string tok = all.substr(pos, end - pos);
            pos = end + 1;
            if (tok == "no_ws") v.no_ws = true;
            else if (tok == "no_gate_fusion") v.no_gate_fusion = true;
            else if (tok == "no_wide_tn") v.no_wide_tn = true;
            else if (tok == "no_slab") v.no_slab = true;
            else if (tok == "fused_fwd") v.fused_fwd = true;
            else if (tok.rfind("dx_reserve=", 0) == 0) {
                const int n = atoi(tok.c_str() + 11);
                if (n >= 0 && n <= 128) v.dx_reserve = n;
            } else if (!tok.empty()) {
                fprintf(stderr, "libaecf_hip: unknown AECF_DEBUG token '%s' (ignored)\n", tok.c_str());
            }
        }
        return v;
    }();
    return e;
}

bool env_no_gate_fusion() { return env_switches().no_gate_fusion; }
bool env_fused_fwd() { return env_switches().fused_fwd; }
int env_dx_reserve() { return env_switches().dx_reserve; }
}  // namespace
namespace aecf {
bool env_no_ws() { return env_switches().no_ws; }
bool env_no_wide_tn() { return env_switches().no_wide_tn; }
bool env_no_slab() { return env_switches().no_slab; }
}  // namespace aecf

extern "C" {

int aecf_pool_forward(const aecf_pool_desc* d, const aecf_pool_fwd_args* a, void* stream) {
    return pool_forward_on(d, a, (hipStream_t)stream);
}

int aecf_pool_backward(const aecf_pool_desc* d, const aecf_pool_bwd_args* a, void* stream) {
    return pool_backward_on(d, a, (hipStream_t)stream);
}

int aecf_curriculum_mask_forward(int64_t rows, int32_t L, int32_t mode, int32_t min_active, float base_mask_prob,
                                 float entropy_target, float eps, const float* weights, const float* uniforms,
                                 float* masked, float* entropy, float* mask_rate, uint8_t* mask_bits, void* stream) {
    if (rows <= 0 || L <= 0) return AECF_ERR_BAD_DIMS;
    if (mode != 1 && mode != 2) return AECF_ERR_BAD_DIMS;
    if (!weights) return AECF_ERR_NULL_POINTER;
    if (mode == 1 && L > 1 && !uniforms) return AECF_ERR_NULL_POINTER;
    if (mode == 1 && L > 64 && !mask_bits) return AECF_ERR_NULL_POINTER;      // (rows beyond 64 keys keep their bits there)
    MaskCfg c = make_mask_cfg(mode, min_active, base_mask_prob, entropy_target, eps, L);
    launch_mask_fwd(rows, L, c, weights, uniforms, masked, entropy, mask_rate, mask_bits, (hipStream_t)stream);
    return launch_status();
}

int aecf_curriculum_mask_backward(int64_t rows, int32_t L, int32_t mode, float eps, const float* weights,
                                  const uint8_t* mask_bits, const float* d_masked, const float* d_entropy,
                                  float* d_weights, void* stream) {
    if (rows <= 0 || L <= 0) return AECF_ERR_BAD_DIMS;
    if (!weights || !d_weights) return AECF_ERR_NULL_POINTER;
    if (mode == 1 && L > 1 && !mask_bits) return AECF_ERR_NULL_POINTER;
    launch_mask_bwd(rows, L, mode, eps, (float)log((double)L), weights, mask_bits, d_masked, d_entropy, d_weights,
                    (hipStream_t)stream);
    return launch_status();
}

size_t aecf_entropy_loss_workspace_bytes(int64_t n) { (void)n; return 1024 * sizeof(float); }

int aecf_entropy_loss_fwd_bwd(int64_t n, int32_t dtype, int32_t last_seq_len, float entropy_target, const void* entropy,
                              float upstream, void* loss, float* d_entropy, void* workspace, void* stream) {
    if (n <= 0) return AECF_ERR_BAD_DIMS;
    if (dtype != AECF_BF16 && dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!entropy || !loss || !workspace) return AECF_ERR_NULL_POINTER;
    const double max_ent = last_seq_len > 1 ? log((double)last_seq_len) : 0.0;   // ref :307
    const float target = (float)(max_ent * (double)entropy_target);
    launch_entropy_loss(dtype, n, target, entropy, upstream, loss, d_entropy, (float*)workspace, (hipStream_t)stream);
    return launch_status();
}

int aecf_sdpa_forward(int64_t B, int32_t S, int32_t T, int32_t E, int32_t dtype, float scale, const void* q,
                      const void* k, const void* v, void* out, float* probs, void* stream) {
    if (B <= 0 || S <= 0 || T <= 0 || E <= 0) return AECF_ERR_BAD_DIMS;
    if (S > 64 || T > 64 || (dtype != AECF_BF16 && dtype != AECF_F32)) return AECF_ERR_UNSUPPORTED;
    if (!q || !k || !v || !out) return AECF_ERR_NULL_POINTER;
    launch_sdpa_fwd(dtype, B, S, T, E, scale, q, k, v, out, probs, (hipStream_t)stream);
    return launch_status();
}

int aecf_sdpa_backward(int64_t B, int32_t S, int32_t T, int32_t E, int32_t dtype, float scale, const void* q,
                       const void* k, const void* v, const float* probs, const void* dout, void* dq, void* dk,
                       void* dv, void* stream) {
    if (B <= 0 || S <= 0 || T <= 0 || E <= 0) return AECF_ERR_BAD_DIMS;
    if (S > 64 || T > 64 || (dtype != AECF_BF16 && dtype != AECF_F32)) return AECF_ERR_UNSUPPORTED;
    if (!q || !k || !v || !probs || !dout || !dq || !dk || !dv) return AECF_ERR_NULL_POINTER;
    launch_sdpa_bwd(dtype, B, S, T, E, scale, q, k, v, probs, dout, dq, dk, dv, (hipStream_t)stream);
    return launch_status();
}

int aecf_philox_uniforms(int64_t n, uint64_t seed, uint64_t offset, uint32_t threads, int64_t element0, float* out, void* stream) {
    if (n <= 0 || threads == 0 || threads % 256 != 0 || offset % 4 != 0 || element0 < 0) return AECF_ERR_BAD_DIMS;
    if (!out) return AECF_ERR_NULL_POINTER;
    PhiloxDraw ph;
    ph.seed = seed; ph.offset = offset; ph.threads = threads; ph.elem0 = (unsigned long long)element0;
    launch_philox_uniforms(n, ph, out, (hipStream_t)stream);
    return launch_status();
}

float aecf_philox_host(uint64_t seed, uint64_t offset, uint32_t threads, int64_t element, uint32_t* raw) {
    PhiloxDraw ph;
    ph.seed = seed; ph.offset = offset; ph.threads = threads ? threads : 256;
    if (raw) {
        const unsigned long long T = ph.threads, per = 4ull * T;
        const unsigned long long it = (unsigned long long)element / per, rem = (unsigned long long)element - it * per;
        const unsigned long long idx = rem - (rem / T) * T, ctr = ph.offset / 4ull + it;
        unsigned int c[4] = {(unsigned int)ctr, (unsigned int)(ctr >> 32), (unsigned int)idx, (unsigned int)(idx >> 32)};
        philox4x32_10(c, (unsigned int)seed, (unsigned int)(seed >> 32));
        for (int i = 0; i < 4; ++i) raw[i] = c[i];
    }
    return philox_uniform_at(ph, element);
}

int aecf_entropy_loss_from_partials(int64_t n, int32_t dtype, const float* partial, void* loss, void* stream) {
    if (n <= 0) return AECF_ERR_BAD_DIMS;
    if (dtype != AECF_BF16 && dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!partial || !loss) return AECF_ERR_NULL_POINTER;
    launch_entropy_from_partials(dtype, n, partial, loss, (hipStream_t)stream);
    return launch_status();
}

int aecf_modality_frontend(int64_t rows, int32_t dim, int32_t dtype, const void* feat, const uint8_t* drop, void* out,
                           uint8_t* present, void* stream) {
    if (rows <= 0 || dim <= 0) return AECF_ERR_BAD_DIMS;
    if (dtype != AECF_BF16 && dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!feat || !present || (drop && !out)) return AECF_ERR_NULL_POINTER;
    launch_modality_frontend(dtype, rows, dim, feat, drop, out, present, (hipStream_t)stream);
    return launch_status();
}

int aecf_loss_fwd_bwd(int64_t rows, int64_t cols, int64_t row_offset, int32_t d, float temperature, float coef, const void* q,
                      const void* k, float* loss_rows, float* dq, float* dk, int64_t n_entropy, int32_t last_seq_len,
                      float entropy_target, const float* entropy, float entropy_upstream, float* entropy_loss,
                      float* d_entropy, void* workspace, size_t workspace_bytes, void* stream) {
    if (rows <= 0 || cols <= 0 || d <= 0 || temperature <= 0.f || n_entropy < 0) return AECF_ERR_BAD_DIMS;
    if (row_offset < 0 || row_offset + rows > cols) return AECF_ERR_BAD_DIMS;
    if (!q || !k || !loss_rows || !dq || !dk || !workspace) return AECF_ERR_NULL_POINTER;
    if (n_entropy > 0 && (!entropy || !entropy_loss)) return AECF_ERR_NULL_POINTER;
    const double max_ent = last_seq_len > 1 ? log((double)last_seq_len) : 0.0;       // ref :301-308
    const float ent_t = (float)(max_ent * (double)entropy_target);
    const float* ent = n_entropy > 0 ? entropy : nullptr;
    if (nce_gemm_supported(AECF_BF16, d, temperature) && workspace_bytes >= nce_gemm_workspace_bytes(rows, cols, d)) {
        launch_nce_gemm_pass1(rows, cols, d, 1.0f / temperature, q, k, workspace, nullptr, (hipStream_t)stream);
        launch_nce_gemm_loss(rows, cols, row_offset, d, 1.0f / temperature, 0, q, k, nullptr, workspace, loss_rows, ent, n_entropy,
                             ent_t, entropy_upstream, d_entropy, entropy_loss, (hipStream_t)stream);
        launch_nce_gemm_grads(rows, cols, row_offset, d, 1.0f / temperature, coef, 0, q, k, workspace, nullptr, 0, dq, dk,
                              (hipStream_t)stream);
        return launch_status();
    }
    if (!nce_flash_supported(AECF_BF16, d)) return AECF_ERR_UNSUPPORTED;
    if (workspace_bytes < nce_flash_workspace_bytes(rows, cols, d)) return AECF_ERR_WORKSPACE;
    launch_nce_flash(rows, cols, row_offset, d, 1.0f / temperature, coef, q, k, loss_rows, dq, dk, workspace, ent, n_entropy, ent_t,
                     entropy_upstream, d_entropy, entropy_loss, (hipStream_t)stream);
    return launch_status();
}

size_t aecf_nce_sym_workspace_bytes(int64_t rows, int64_t cols, int32_t d) {
    if (rows <= 0 || cols <= 0 || d <= 0 || d % 64 != 0) return 0;
    return nce_gemm_workspace_bytes(rows, cols, d);
}

int aecf_nce_sym_pass1(int64_t rows, int64_t cols, int32_t d, float temperature, const void* a, const void* b, void* workspace,
                       size_t workspace_bytes, float* col_sums, void* stream) {
    if (rows <= 0 || cols <= 0 || d <= 0 || temperature <= 0.f || rows > cols) return AECF_ERR_BAD_DIMS;
    if (!nce_gemm_supported(AECF_BF16, d, temperature)) return AECF_ERR_UNSUPPORTED;
    if (!a || !b || !workspace || !col_sums) return AECF_ERR_NULL_POINTER;
    if (workspace_bytes < nce_gemm_workspace_bytes(rows, cols, d)) return AECF_ERR_WORKSPACE;
    launch_nce_gemm_pass1(rows, cols, d, 1.0f / temperature, a, b, workspace, col_sums, (hipStream_t)stream);
    return launch_status();
}

int aecf_nce_sym_loss(int64_t rows, int64_t cols, int64_t row_offset, int32_t d, float temperature, const void* a, const void* b,
                      const float* col_sums, void* workspace, size_t workspace_bytes, float* loss_rows, int64_t n_entropy,
                      int32_t last_seq_len, float entropy_target, const float* entropy, float entropy_upstream, float* entropy_loss,
                      float* d_entropy, void* stream) {
    if (rows <= 0 || cols <= 0 || d <= 0 || temperature <= 0.f || n_entropy < 0) return AECF_ERR_BAD_DIMS;
    if (row_offset < 0 || row_offset + rows > cols) return AECF_ERR_BAD_DIMS;
    if (!nce_gemm_supported(AECF_BF16, d, temperature)) return AECF_ERR_UNSUPPORTED;
    if (!a || !b || !col_sums || !workspace || !loss_rows) return AECF_ERR_NULL_POINTER;
    if (n_entropy > 0 && (!entropy || !entropy_loss)) return AECF_ERR_NULL_POINTER;
    if (workspace_bytes < nce_gemm_workspace_bytes(rows, cols, d)) return AECF_ERR_WORKSPACE;
    const double max_ent = last_seq_len > 1 ? log((double)last_seq_len) : 0.0;       // ref :301-308
    launch_nce_gemm_loss(rows, cols, row_offset, d, 1.0f / temperature, 1, a, b, col_sums, workspace, loss_rows,
                         n_entropy > 0 ? entropy : nullptr, n_entropy, (float)(max_ent * (double)entropy_target), entropy_upstream,
                         d_entropy, entropy_loss, (hipStream_t)stream);
    return launch_status();
}

int aecf_nce_sym_grads(int64_t rows, int64_t cols, int64_t row_offset, int32_t d, float temperature, float coef, const void* a,
                       const void* b, void* workspace, size_t workspace_bytes, const float* upstream, int32_t grad_dtype, void* da,
                       void* db, void* stream) {
    if (rows <= 0 || cols <= 0 || d <= 0 || temperature <= 0.f) return AECF_ERR_BAD_DIMS;
    if (row_offset < 0 || row_offset + rows > cols) return AECF_ERR_BAD_DIMS;
    if (!nce_gemm_supported(AECF_BF16, d, temperature)) return AECF_ERR_UNSUPPORTED;
    if (grad_dtype != AECF_BF16 && grad_dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!a || !b || !workspace || !da || !db) return AECF_ERR_NULL_POINTER;
    if (workspace_bytes < nce_gemm_workspace_bytes(rows, cols, d)) return AECF_ERR_WORKSPACE;
    launch_nce_gemm_grads(rows, cols, row_offset, d, 1.0f / temperature, coef, 1, a, b, workspace, upstream,
                          grad_dtype == AECF_BF16 ? 1 : 0, da, db, (hipStream_t)stream);
    return launch_status();
}

int aecf_route_build(int64_t rows, const uint8_t* present_a, const uint8_t* present_b, int32_t* route, int32_t* slot,
                     int32_t* index, int32_t* counts, void* stream) {
    if (rows <= 0 || rows > 0x7fffffff) return AECF_ERR_BAD_DIMS;
    if (!present_a || !present_b || !route || !slot || !index || !counts) return AECF_ERR_NULL_POINTER;
    launch_route_build(rows, present_a, present_b, route, slot, index, counts, (hipStream_t)stream);
    return launch_status();
}

int aecf_rows_gather(int32_t njobs, const void* const* src, const int64_t* src_pitch, const int32_t* const* index,
                     const int64_t* n, void* const* dst, const int64_t* dst_pitch, int64_t row_bytes, void* stream) {
    if (njobs < 1 || njobs > 3 || row_bytes <= 0 || row_bytes % 2 != 0) return AECF_ERR_BAD_DIMS;
    if (!src || !src_pitch || !index || !n || !dst || !dst_pitch) return AECF_ERR_NULL_POINTER;
    for (int k = 0; k < njobs; ++k) {
        if (n[k] < 0 || src_pitch[k] % 2 != 0 || dst_pitch[k] % 2 != 0) return AECF_ERR_BAD_DIMS;
        if (n[k] > 0 && (!src[k] || !index[k] || !dst[k])) return AECF_ERR_NULL_POINTER;
    }
    launch_rows_gather(njobs, src, src_pitch, index, n, dst, dst_pitch, row_bytes, (hipStream_t)stream);
    return launch_status();
}

int aecf_rows_select(int64_t rows, int64_t row_bytes, const int32_t* route, const int32_t* slot, const void* const* src,
                     const int64_t* src_pitch, void* dst, int64_t dst_pitch, void* stream) {
    if (rows <= 0 || row_bytes <= 0 || row_bytes % 2 != 0 || dst_pitch % 2 != 0) return AECF_ERR_BAD_DIMS;
    if (!route || !src || !src_pitch || !dst) return AECF_ERR_NULL_POINTER;          // slot may be NULL: identity
    for (int k = 0; k < 3; ++k)
        if (src_pitch[k] % 2 != 0) return AECF_ERR_BAD_DIMS;
    launch_rows_select(rows, row_bytes, route, slot, src, src_pitch, dst, dst_pitch, (hipStream_t)stream);
    return launch_status();
}

int aecf_cast_f32_to_bf16(int32_t n, const float* const* src, void* const* dst, const int64_t* numel, void* stream) {
    if (n < 0 || n > 8) return AECF_ERR_BAD_DIMS;
    if (n == 0) return AECF_OK;
    if (!src || !dst || !numel) return AECF_ERR_NULL_POINTER;
    for (int i = 0; i < n; ++i) {
        if (numel[i] < 0) return AECF_ERR_BAD_DIMS;
        if (numel[i] > 0 && (!src[i] || !dst[i])) return AECF_ERR_NULL_POINTER;
    }
    launch_cast_f32_bf16_multi(n, src, dst, numel, (hipStream_t)stream);
    return launch_status();
}

int aecf_adamw_step(int32_t n, void* const* param, const void* const* grad, void* const* exp_avg, void* const* exp_avg_sq,
                    void* const* step, const int64_t* numel, void* ticket, float lr, float beta1, float beta2, float eps,
                    float weight_decay, void* stream) {
    if (n <= 0) return AECF_ERR_BAD_DIMS;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step || !numel || !ticket) return AECF_ERR_NULL_POINTER;
    for (int i = 0; i < n; ++i) {
        if (numel[i] <= 0) return AECF_ERR_BAD_DIMS;
        if (!param[i] || !grad[i] || !exp_avg[i] || !exp_avg_sq[i] || !step[i]) return AECF_ERR_NULL_POINTER;
    }
    launch_adamw_multi(n, (float* const*)param, (const float* const*)grad, (float* const*)exp_avg, (float* const*)exp_avg_sq,
                       (float* const*)step, numel, (unsigned int*)ticket, lr, beta1, beta2, eps, weight_decay, (hipStream_t)stream);
    return launch_status();
}

int aecf_rows_split(int64_t rows, int64_t row_bytes, const int32_t* route, const void* src, void* const* dst, void* stream) {
    if (rows <= 0 || row_bytes <= 0 || row_bytes % 2 != 0) return AECF_ERR_BAD_DIMS;
    if (!route || !src || !dst) return AECF_ERR_NULL_POINTER;
    launch_rows_split(rows, row_bytes, route, src, dst, (hipStream_t)stream);
    return launch_status();
}

int aecf_front_pair(int64_t rows, int32_t dim_a, int32_t dim_b, int32_t dtype, const void* feat_a, const void* feat_b,
                    const float* uniforms, float missing_prob, const uint8_t* drop_a, const uint8_t* drop_b, void* out_a,
                    void* out_b, uint8_t* present_a, uint8_t* present_b, int32_t* cls, void* stream) {
    if (rows <= 0 || dim_a <= 0 || dim_b <= 0) return AECF_ERR_BAD_DIMS;
    if (dtype != AECF_BF16 && dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!feat_a || !feat_b || !out_a || !out_b || !present_a || !present_b || !cls) return AECF_ERR_NULL_POINTER;
    if (uniforms && (drop_a || drop_b)) return AECF_ERR_BAD_DIMS;                      // one source of decisions
    if (out_a == feat_a || out_b == feat_b) return AECF_ERR_UNSUPPORTED;               // (the row is read twice)
    launch_front_pair(dtype, rows, dim_a, dim_b, feat_a, feat_b, uniforms, missing_prob, drop_a, drop_b, out_a, out_b,
                      present_a, present_b, cls, (hipStream_t)stream);
    return launch_status();
}

int aecf_l2norm_forward(int64_t n, int32_t d, int32_t dtype, float eps, const void* z, void* zn, float* inv_norm,
                        void* stream) {
    if (n <= 0 || d <= 0) return AECF_ERR_BAD_DIMS;
    if (dtype != AECF_BF16 && dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!z || !zn || !inv_norm) return AECF_ERR_NULL_POINTER;
    launch_l2norm_fwd(dtype, n, d, eps, z, zn, inv_norm, (hipStream_t)stream);
    return launch_status();
}

int aecf_l2norm_backward(int64_t n, int32_t d, int32_t dtype, const void* zn, const float* inv_norm, const float* dzn,
                         void* dz, void* stream) {
    if (n <= 0 || d <= 0) return AECF_ERR_BAD_DIMS;
    if (dtype != AECF_BF16 && dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!zn || !inv_norm || !dzn || !dz) return AECF_ERR_NULL_POINTER;
    launch_l2norm_bwd(dtype, n, d, zn, inv_norm, dzn, dz, (hipStream_t)stream);
    return launch_status();
}

// bytes the generic (logits-in-memory) form of aecf_nce_fwd_bwd carves from its workspace
static size_t nce_generic_workspace_bytes(int64_t rows, int64_t cols, int32_t d, int32_t dtype) {
    const size_t es = esize(dtype);
    return align_up((size_t)rows * cols * 4) + align_up((size_t)rows * cols * es) + align_up((size_t)d * cols * es);
}

// The form aecf_nce_fwd_bwd takes depends on the TEMPERATURE as well (the tile GEMMs' constant-shift softmax needs
// 1/T within the float32 exponent range), which this query does not see: it answers the LARGEST workspace any form
// that can be selected for (d, dtype) needs, so the call never runs a form on a buffer sized for another one.
size_t aecf_nce_workspace_bytes(int64_t rows, int64_t cols, int32_t d, int32_t dtype) {
    if (rows <= 0 || cols <= 0 || d <= 0) return 0;
    size_t need = 0;
    if (nce_gemm_supported(dtype, d, 1.0f)) need = nce_gemm_workspace_bytes(rows, cols, d);
    if (nce_flash_supported(dtype, d)) {
        const size_t f = nce_flash_workspace_bytes(rows, cols, d);      // (taken when the tile form refuses the temperature)
        return f > need ? f : need;
    }
    if ((d % 64 == 0 && cols % 64 == 0) || need == 0) {
        const size_t g = nce_generic_workspace_bytes(rows, cols, d, dtype);
        if (g > need) need = g;
    }
    return need;
}

size_t aecf_nce_stream_workspace_bytes(int64_t rows, int64_t cols, int32_t d, int32_t dtype) {
    if (rows <= 0 || cols <= 0 || d <= 0 || !nce_flash_supported(dtype, d)) return 0;
    return nce_flash_workspace_bytes(rows, cols, d);
}

int aecf_nce_fwd_bwd(int64_t rows, int64_t cols, int64_t row_offset, int32_t d, int32_t dtype, float temperature,
                     float coef, const void* q, const void* k, float* loss_rows, float* dq, float* dk, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (rows <= 0 || cols <= 0 || d <= 0 || temperature <= 0.f) return AECF_ERR_BAD_DIMS;
    if (row_offset < 0 || row_offset + rows > cols) return AECF_ERR_BAD_DIMS;
    if (dtype != AECF_BF16 && dtype != AECF_F32) return AECF_ERR_UNSUPPORTED;
    if (!q || !k || !loss_rows || !dq || !dk || !workspace) return AECF_ERR_NULL_POINTER;
    if (nce_gemm_supported(dtype, d, temperature) && workspace_bytes >= nce_gemm_workspace_bytes(rows, cols, d)) {
        launch_nce_gemm_pass1(rows, cols, d, 1.0f / temperature, q, k, workspace, nullptr, (hipStream_t)stream);
        launch_nce_gemm_loss(rows, cols, row_offset, d, 1.0f / temperature, 0, q, k, nullptr, workspace, loss_rows, nullptr, 0, 0.f,
                             0.f, nullptr, nullptr, (hipStream_t)stream);
        launch_nce_gemm_grads(rows, cols, row_offset, d, 1.0f / temperature, coef, 0, q, k, workspace, nullptr, 0, dq, dk,
                              (hipStream_t)stream);
        return launch_status();
    }
    if (nce_flash_supported(dtype, d)) {         // streaming form: any rows / cols, workspace O(rows d)
        if (workspace_bytes < nce_flash_workspace_bytes(rows, cols, d)) return AECF_ERR_WORKSPACE;
        launch_nce_flash(rows, cols, row_offset, d, 1.0f / temperature, coef, q, k, loss_rows, dq, dk, workspace, nullptr, 0,
                         0.f, 0.f, nullptr, nullptr, (hipStream_t)stream);
        return launch_status();
    }
    if (d % 64 != 0 || cols % 64 != 0) return AECF_ERR_UNSUPPORTED;
    if (workspace_bytes < nce_generic_workspace_bytes(rows, cols, d, dtype)) return AECF_ERR_WORKSPACE;   // its OWN size
    hipStream_t s = (hipStream_t)stream;
    const size_t es = esize(dtype);
    char* ws = (char*)workspace;
    float* S = (float*)ws;
    void* G = ws + align_up((size_t)rows * cols * 4);
    void* kT = (char*)G + align_up((size_t)rows * cols * es);

    GemmNtArgs g1;       // S = q k^T  (float32 logits before the temperature)
    g1.a = q; g1.w = k; g1.bias = nullptr; g1.c = S; g1.probs = nullptr; g1.R = rows; g1.N = (int)cols; g1.K = d;
    g1.lda = d; g1.M = 1; g1.H = 1; g1.hd = d; g1.pooled = 0; g1.out_f32 = 1; g1.v_out = nullptr;
    launch_gemm_nt(dtype, g1, s);
    launch_nce_rows(dtype, rows, cols, row_offset, 1.0f / temperature, coef, S, G, loss_rows, s);
    launch_transpose_rect(dtype, k, kT, cols, d, s);
    GemmNtArgs g2;       // dq = G k
    g2.a = G; g2.w = kT; g2.bias = nullptr; g2.c = dq; g2.probs = nullptr; g2.R = rows; g2.N = d; g2.K = (int)cols;
    g2.lda = cols; g2.M = 1; g2.H = 1; g2.hd = d; g2.pooled = 0; g2.out_f32 = 1; g2.v_out = nullptr;
    launch_gemm_nt(dtype, g2, s);
    GemmTnArgs t;        // dk = G^T q   (reduction over the local rows)
    t.lhs = G; t.rhs = q; t.probs = nullptr; t.dsbuf = nullptr; t.out = dk; t.colsum = nullptr; t.u = nullptr;
    t.B = rows; t.M = 1; t.E = d; t.H = 1; t.hd = d; t.Ej = (int)cols; t.splits = 1; t.u_splits = 0; t.u_rows_per_split = 0;
    t.rows_per_split = (rows + 63) / 64 * 64; t.pooled = 0;
    launch_gemm_tn(dtype, t, s);
    return launch_status();
}

}  // extern "C"
