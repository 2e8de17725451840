// Presence routing around the fusion pool (the caller's side of ref
// xrays/train_xrays_example.py:205-234).  The reference routes rows with boolean masks, torch.where, torch.stack and
// index_put (six host synchronisations and ~15 launches per step).  Here the routing is a table built on the device:
//   route_build  : one pass over the two presence vectors -> class of every row (both / only-a / only-b / none), its
//                  slot inside its class (stable order = ascending row index, as torch.where gives), the inverse lists
//                  and the four class sizes (the only values the host reads back: one 16-byte copy);
//   rows_gather  : up to three jobs "dst[i] = src[index[i]]" in one launch (the [n_both, 2, E] pool input from the two
//                  encoder outputs; the single-modality inputs of the two side projections; the backward of rows_select);
//   rows_select  : dst[b] = src_{class(b)}[slot(b)] or zeros, every row of dst written exactly once (the fused [B, 2E]
//                  rows from the three projected branches; the backward of rows_gather) -- no memset, no index_put.
// Rows are moved as bytes (16-byte pieces when pitch and width allow it), one wavefront per row.
#include "aecf_kernels.h"

namespace aecf {

namespace {

__global__ __launch_bounds__(1024) void route_build_kernel(int64_t rows, const uint8_t* __restrict__ pa,
                                                           const uint8_t* __restrict__ pb, int32_t* __restrict__ route,
                                                           int32_t* __restrict__ slot, int32_t* __restrict__ index,
                                                           int32_t* __restrict__ counts) {
    __shared__ int wcnt[3][16];
    __shared__ int base[4];
    const int lane = lane_id(), w = wave_id();
    if (threadIdx.x < 4) base[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t r0 = 0; r0 < rows; r0 += 1024) {
        const int64_t r = r0 + threadIdx.x;
        int cls = 4;                                       // 4 = past the end
        if (r < rows) {
            const bool a = pa[r] != 0, b = pb[r] != 0;
            cls = a ? (b ? 0 : 1) : (b ? 2 : 3);
        }
        int rank[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned long long m = __ballot(cls == c);
            rank[c] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[c][w] = __popcll(m);
        }
        __syncthreads();
        if (cls < 3) {
            int off = base[cls] + rank[cls];
            for (int ww = 0; ww < w; ++ww) off += wcnt[cls][ww];
            slot[r] = off;
            index[(int64_t)cls * rows + off] = (int32_t)r;
        } else if (cls == 3) {
            slot[r] = 0;
        }
        if (cls < 4) route[r] = cls;
        int none = __popcll(__ballot(cls == 3));
        __syncthreads();
        if (threadIdx.x < 3) {
            int t = 0;
            for (int ww = 0; ww < 16; ++ww) t += wcnt[threadIdx.x][ww];
            base[threadIdx.x] += t;
        }
        if (lane == 0 && none) atomicAdd(&base[3], none);
        __syncthreads();
    }
    if (threadIdx.x < 4) counts[threadIdx.x] = base[threadIdx.x];
}

struct GatherJobs {
    const char* src[3];
    const int32_t* index[3];
    char* dst[3];
    int64_t src_pitch[3], dst_pitch[3], n[3];
    int64_t row_bytes;
    int njobs;
};

__device__ __forceinline__ void copy_row(const char* __restrict__ s, char* __restrict__ d, int64_t bytes, int lane) {
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | (uintptr_t)bytes) & 15) == 0) {
        for (int64_t o = (int64_t)lane * 16; o < bytes; o += 64 * 16)
            *reinterpret_cast<u32x4*>(d + o) = *reinterpret_cast<const u32x4*>(s + o);
    } else {
        for (int64_t o = (int64_t)lane * 2; o < bytes; o += 64 * 2)
            *reinterpret_cast<unsigned short*>(d + o) = *reinterpret_cast<const unsigned short*>(s + o);
    }
}
__device__ __forceinline__ void zero_row(char* __restrict__ d, int64_t bytes, int lane) {
    if (((reinterpret_cast<uintptr_t>(d) | (uintptr_t)bytes) & 15) == 0) {
        for (int64_t o = (int64_t)lane * 16; o < bytes; o += 64 * 16) *reinterpret_cast<u32x4*>(d + o) = u32x4{0u, 0u, 0u, 0u};
    } else {
        for (int64_t o = (int64_t)lane * 2; o < bytes; o += 64 * 2) *reinterpret_cast<unsigned short*>(d + o) = 0;
    }
}

__global__ __launch_bounds__(256) void rows_gather_kernel(GatherJobs j) {
    int64_t r = (int64_t)blockIdx.x * 4 + wave_id();
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k >= j.njobs) return;
        if (r < j.n[k]) {
            const int64_t s = j.index[k][r];
            copy_row(j.src[k] + s * j.src_pitch[k], j.dst[k] + r * j.dst_pitch[k], j.row_bytes, lane);
            return;
        }
        r -= j.n[k];
    }
}

struct SelectArgs {
    const char* src[3];
    int64_t src_pitch[3];
    const int32_t* route;
    const int32_t* slot;
    char* dst;
    int64_t dst_pitch, row_bytes, rows;
};

__global__ __launch_bounds__(256) void rows_select_kernel(SelectArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 4 + wave_id();
    if (r >= a.rows) return;
    const int lane = lane_id();
    const int c = a.route[r];
    char* d = a.dst + r * a.dst_pitch;
    const char* s = nullptr;
    if (c == 0) s = a.src[0]; else if (c == 1) s = a.src[1]; else if (c == 2) s = a.src[2];
    if (s) {
        const int64_t p = c == 0 ? a.src_pitch[0] : (c == 1 ? a.src_pitch[1] : a.src_pitch[2]);
        copy_row(s + (a.slot ? (int64_t)a.slot[r] : r) * p, d, a.row_bytes, lane);
    } else {
        zero_row(d, a.row_bytes, lane);
    }
}

// dst_c[r] = (route[r] == c) ? src[r] : 0 for c = 0..2: the backward of rows_select with identity slots (static routing: every
// branch keeps all rows, a row's gradient goes to the branch that owned it), one launch for the three branches
struct SplitArgs {
    const char* src;
    char* dst[3];
    const int32_t* route;
    int64_t row_bytes, rows;
};
__global__ __launch_bounds__(256) void rows_split_kernel(SplitArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 4 + wave_id();
    if (r >= a.rows) return;
    const int lane = lane_id();
    const int c = a.route[r];
    const char* s = a.src + r * a.row_bytes;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!a.dst[k]) continue;
        char* d = a.dst[k] + r * a.row_bytes;
        if (c == k) copy_row(s, d, a.row_bytes, lane); else zero_row(d, a.row_bytes, lane);
    }
}

// Both modality front-ends, the missing-modality decisions and the row classes of one batch in ONE launch (static routing:
// ref xrays/train_xrays_example.py:156-177 draws, :173-176 zeroing, :202-207 presence and the three masks).  One wave per row.
//   decisions: from uniforms [3, rows] (modality a dropped if u0 < p, b if u1 < p, a row that would lose both keeps a when
//   u2 > 0.5 else b -- AECFModel.draw_missing's rule) or from the caller's drop vectors, or none;
//   present_x = not dropped and ||row|| > 1e-6 (float32 sum of squares; false for NaN rows);
//   out_x = row if present_x else zeros;  cls = 0 both, 1 only a, 2 only b, 3 neither.
struct FrontPairArgs {
    const void* feat[2];
    void* out[2];
    uint8_t* present[2];
    const uint8_t* drop[2];
    const float* uniforms;
    int32_t* cls;
    int64_t rows;
    int dim[2];
    float missing_prob;
};
template <typename T>
__global__ __launch_bounds__(256) void front_pair_kernel(FrontPairArgs a) {
    using X = Tr<T>;
    typedef typename X::elem elem;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave_id();
    if (r >= a.rows) return;
    const int lane = lane_id();
    bool dropped[2] = {false, false};
    if (a.uniforms) {
        const float u0 = a.uniforms[r], u1 = a.uniforms[a.rows + r], u2 = a.uniforms[2 * a.rows + r];
        const bool da = u0 < a.missing_prob, db = u1 < a.missing_prob, clash = da && db, keep_a = u2 > 0.5f;
        dropped[0] = da && !(clash && keep_a);
        dropped[1] = db && !(clash && !keep_a);
    } else {
        dropped[0] = a.drop[0] && a.drop[0][r] != 0;
        dropped[1] = a.drop[1] && a.drop[1][r] != 0;
    }
    bool here[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int dim = a.dim[k];
        const elem* src = reinterpret_cast<const elem*>(a.feat[k]) + r * dim;
        elem* dst = reinterpret_cast<elem*>(a.out[k]) + r * dim;
        const bool vec = (dim % X::EPL == 0) && ((reinterpret_cast<uintptr_t>(a.feat[k]) | reinterpret_cast<uintptr_t>(a.out[k])) % 16 == 0);
        float ss = 0.f;
        if (!dropped[k]) {
            if (vec) {
                for (int c = lane * X::EPL; c < dim; c += 64 * X::EPL) {
                    float f[X::EPL];
                    X::unpack(X::load(src + c), f);
#pragma unroll
                    for (int e = 0; e < X::EPL; ++e) ss = fmaf(f[e], f[e], ss);
                }
            } else {
                for (int c = lane; c < dim; c += 64) { const float f = X::to_f32(src[c]); ss = fmaf(f, f, ss); }
            }
        }
        ss = reduce_wave(ss);
        here[k] = !dropped[k] && sqrtf(ss) > 1e-6f;
        if (vec) {
            for (int c = lane * X::EPL; c < dim; c += 64 * X::EPL)
                *reinterpret_cast<typename X::frag*>(dst + c) = here[k] ? X::load(src + c) : X::zero();
        } else {
            for (int c = lane; c < dim; c += 64) dst[c] = here[k] ? src[c] : X::from_f32(0.f);
        }
    }
    if (lane == 0) {
        a.present[0][r] = here[0] ? 1 : 0;
        a.present[1][r] = here[1] ? 1 : 0;
        a.cls[r] = here[0] ? (here[1] ? 0 : 1) : (here[1] ? 2 : 3);
    }
}

}  // namespace

void launch_rows_split(int64_t rows, int64_t row_bytes, const int32_t* route, const void* src, void* const* dst, hipStream_t s) {
    SplitArgs a;
    a.src = (const char*)src;
    for (int k = 0; k < 3; ++k) a.dst[k] = (char*)dst[k];
    a.route = route;
    a.row_bytes = row_bytes;
    a.rows = rows;
    rows_split_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>(a);
}

void launch_front_pair(int dtype, int64_t rows, int dim_a, int dim_b, const void* feat_a, const void* feat_b, const float* uniforms,
                       float missing_prob, const uint8_t* drop_a, const uint8_t* drop_b, void* out_a, void* out_b,
                       uint8_t* present_a, uint8_t* present_b, int32_t* cls, hipStream_t s) {
    FrontPairArgs a;
    a.feat[0] = feat_a; a.feat[1] = feat_b;
    a.out[0] = out_a; a.out[1] = out_b;
    a.present[0] = present_a; a.present[1] = present_b;
    a.drop[0] = drop_a; a.drop[1] = drop_b;
    a.uniforms = uniforms;
    a.cls = cls;
    a.rows = rows;
    a.dim[0] = dim_a; a.dim[1] = dim_b;
    a.missing_prob = missing_prob;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == 0) front_pair_kernel<BF16><<<grid, block, 0, s>>>(a); else front_pair_kernel<F32><<<grid, block, 0, s>>>(a);
}

void launch_route_build(int64_t rows, const uint8_t* pa, const uint8_t* pb, int32_t* route, int32_t* slot, int32_t* index,
                        int32_t* counts, hipStream_t s) {
    route_build_kernel<<<dim3(1), dim3(1024), 0, s>>>(rows, pa, pb, route, slot, index, counts);
}

void launch_rows_gather(int njobs, const void* const* src, const int64_t* src_pitch, const int32_t* const* index,
                        const int64_t* n, void* const* dst, const int64_t* dst_pitch, int64_t row_bytes, hipStream_t s) {
    GatherJobs j;
    int64_t total = 0;
    for (int k = 0; k < 3; ++k) {
        const bool on = k < njobs;
        j.src[k] = on ? (const char*)src[k] : nullptr;
        j.index[k] = on ? index[k] : nullptr;
        j.dst[k] = on ? (char*)dst[k] : nullptr;
        j.src_pitch[k] = on ? src_pitch[k] : 0;
        j.dst_pitch[k] = on ? dst_pitch[k] : 0;
        j.n[k] = on ? n[k] : 0;
        total += j.n[k];
    }
    j.row_bytes = row_bytes;
    j.njobs = njobs;
    if (total == 0) return;
    rows_gather_kernel<<<dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s>>>(j);
}

void launch_rows_select(int64_t rows, int64_t row_bytes, const int32_t* route, const int32_t* slot, const void* const* src,
                        const int64_t* src_pitch, void* dst, int64_t dst_pitch, hipStream_t s) {
    SelectArgs a;
    for (int k = 0; k < 3; ++k) { a.src[k] = (const char*)src[k]; a.src_pitch[k] = src_pitch[k]; }
    a.route = route; a.slot = slot; a.dst = (char*)dst; a.dst_pitch = dst_pitch; a.row_bytes = row_bytes; a.rows = rows;
    rows_select_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>(a);
}

}  // namespace aecf
