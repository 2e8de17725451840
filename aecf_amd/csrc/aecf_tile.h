// LDS tile engine shared by the MFMA kernels (gfx950).
//
// An LDS tile is ROWS x 128 bytes: one row = the K-slice of one matrix row (64 bf16 or 32 f32 = two MFMA
// K-steps).  The 8 16-byte chunks of a row are XOR-swizzled with s(row) = ((row >> 1) ^ (row >> 4)) & 7:
//   * fragment read (ds_read_b128, 16 consecutive rows, same logical chunk): (row & 1, chunk ^ s) covers the
//     16 distinct 16-byte slots of the 256-byte bank row -> conflict-free;
//   * direct staging write (8 consecutive lanes = one row, chunks 0..7): 8 distinct slots;
//   * transposed staging write (8 consecutive lanes = rows 8*f + i, one chunk): s takes 8 distinct values.
// Tiles are filled from registers (coalesced 16-byte global loads issued one tile ahead, written after the
// barrier), so the staging can also transform the data on the way in (transpose for batch-reduction GEMMs,
// probability-weighted pooling over modalities).
#pragma once
#include "aecf_common.h"

namespace aecf {

constexpr int TILE_ROW_BYTES = 128;

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * TILE_ROW_BYTES + ((chunk ^ (((row >> 1) ^ (row >> 4)) & 7)) << 4);
}

// XCD-aware block -> (row panel, column tile) map for [rows x cols] tilings whose column tiles share the row panel's operand.
// Blocks are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2 and its 32 CUs), so XCD x is given the CONTIGUOUS
// run [x per, (x + 1) per) of the (panel, column) pairs, per = ceil(npanel ncol / 8): the column tiles of a panel sit on one XCD,
// adjacent in time -- the panel is fetched from HBM once and re-read from that XCD's L2 -- and every XCD gets the same number
// of blocks.  (Until round 5 whole panels were dealt to the XCDs, panel p to XCD p % 8: with 42 panels of 6 column groups --
// d = 768 -- two XCDs got 36 one-per-CU blocks for their 32 CUs and the launch took two rounds there.)  Launch with
// xcd_grid(npanel, ncol) blocks; returns false for padding ids.
__host__ __device__ __forceinline__ unsigned int xcd_grid(unsigned int npanel, unsigned int ncol) {
    return ((npanel * ncol + 7u) / 8u) * 8u;
}
__device__ __forceinline__ bool xcd_tile(unsigned int id, unsigned int npanel, unsigned int ncol, unsigned int& panel,
                                         unsigned int& col) {
    const unsigned int total = npanel * ncol, per = (total + 7u) / 8u;
    const unsigned int q = (id & 7u) * per + (id >> 3);
    panel = q / ncol;
    col = q - panel * ncol;
    return (id >> 3) < per && q < total;
}

template <typename T> struct TileK;   // elements of K per 128-byte LDS row
template <> struct TileK<BF16> { static constexpr int value = 64; };
template <> struct TileK<F32> { static constexpr int value = 32; };

// Cooperative register staging of a ROWS x 128 B tile from a row-major source, NT threads.
// chunk c = tid + NT*i  ->  row = c >> 3, 16-byte chunk = c & 7: 8 consecutive lanes read one full 128-B line.
template <int ROWS, int NT>
struct DirectStage {
    static constexpr int N = (ROWS * 8 + NT - 1) / NT;
    u32x4 r[N];
    // src: byte pointer to element (row0, k-byte offset) of the source; ld_bytes: row pitch; rows_valid: rows that
    // exist; chunks_valid: 16-byte chunks of the K-slice that exist (the rest is zero-filled)
    __device__ __forceinline__ void load(const char* __restrict__ src, int64_t ld_bytes, int rows_valid,
                                         int chunks_valid = 8) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = threadIdx.x + NT * i;
            const int row = c >> 3, ch = c & 7;
            r[i] = (row < rows_valid && ch < chunks_valid)
                       ? *reinterpret_cast<const u32x4*>(src + row * ld_bytes + ch * 16)
                       : u32x4{0u, 0u, 0u, 0u};
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = threadIdx.x + NT * i;
            if (ROWS * 8 % NT == 0 || c < ROWS * 8)
                *reinterpret_cast<u32x4*>(lds + lds_off(c >> 3, c & 7)) = r[i];
        }
    }
};

// LDS-DMA staging of a ROWS x 128 B tile (global_load_lds_dwordx4: no VGPR round trip, no ds_write).  The LDS
// destination of one wave-instruction is linear (wave-uniform base + lane * 16), which is exactly this tile
// format's physical chunk order (chunk c = tid + NT*i at byte 16*c); the XOR swizzle therefore goes on the per-lane
// SOURCE address: physical chunk ph of row r holds logical chunk ph ^ s(r).  Rows past rows_valid re-read the last
// valid row (their products are never stored).  Completion: the DMA counts on vmcnt; hipcc drains it in front
// of the next __syncthreads(), so "barrier; issue next tile; compute this tile" overlaps one tile of loads.
typedef __attribute__((address_space(3))) void lds_void_t;
template <int ROWS, int NT>
__device__ __forceinline__ void dma_tile(const char* __restrict__ src, int64_t ld_bytes, int rows_valid, char* lds) {
    constexpr int N = ROWS * 8 / NT;
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = threadIdx.x + NT * i;
        const int row = c >> 3, ph = c & 7;
        const int rowc = row < rows_valid ? row : rows_valid - 1;
        const int logical = ph ^ (((row >> 1) ^ (row >> 4)) & 7);
        __builtin_amdgcn_global_load_lds(src + rowc * ld_bytes + logical * 16,
                                         (lds_void_t*)(lds + (wbase + NT * i) * 16), 16, 0, 0);
    }
}

// ---- row tiles of the weight-stationary / streaming kernels: NROWS rows of 64 KT bytes, 16-byte chunk p of row r stored
// at chunk p ^ key(r), key(r) = (r / KEYDIV) & 15 (conflict-free ds_read_b128 fragment reads of 16 consecutive rows) ----
// LDS-DMA of such a tile issued through inline asm: invisible to hipcc's waitcnt insertion, which otherwise puts s_waitcnt vmcnt(0)
// in front of the first LDS read that follows a global_load_lds it can see whenever ordinary loads are in flight as well
// (the copy would then be serialised with the compute of the step it is meant to fly behind).  The caller retires it with
// its own s_waitcnt vmcnt(0).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int KT, int NROWS, int KEYDIV, int NT = 512>
__device__ __forceinline__ void ws_dma_rows_asm(const char* __restrict__ src, unsigned int ld_bytes, int rows_valid, char* lds) {
    constexpr int CPR = 4 * KT;
    constexpr int TOTAL = NROWS * CPR;
    constexpr int NI = (TOTAL + NT - 1) / NT;
    static_assert(TOTAL % 64 == 0, "whole wave-instructions");
    const int wbase = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (wbase + NT * i < TOTAL) {                 // wave-uniform
            const int c = threadIdx.x + NT * i;
            const int row = c / CPR, p = c - row * CPR;
            const int rowc = row < rows_valid ? row : rows_valid - 1;
            const int key = (row / KEYDIV) & 15;
            const unsigned int voff = (unsigned)rowc * ld_bytes + (unsigned)((p ^ key) << 4);
            const unsigned int dst = (unsigned)(size_t)(lds_void_t*)(lds + (wbase + NT * i) * 16);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst)
                         : "memory", "m0");
        }
    }
}
#pragma clang diagnostic pop

// MFMA A/B operand of 8 consecutive ROWS (the K index) at one column per lane, from a row-major LDS tile, by two
// transposed reads (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block, lane i receives column i)
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t lds_v4i16_t;
__device__ __forceinline__ u32x4 tr_frag16(const char* tile, int addr_lo, int addr_hi) {
    const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)(tile + addr_lo));
    const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)(tile + addr_hi));
    const u32x2 l = __builtin_bit_cast(u32x2, lo), h = __builtin_bit_cast(u32x2, hi);
    return u32x4{l[0], l[1], h[0], h[1]};
}

template <typename T>
__device__ __forceinline__ typename Tr<T>::frag lds_frag(const char* lds, int row, int chunk) {
    return *reinterpret_cast<const typename Tr<T>::frag*>(lds + lds_off(row, chunk));
}

// acc[rt][ct] += A(rows rowA0 + 16 rt ..) * B(rows rowB0 + 16 ct ..)^T over the two K-steps of the tile
template <typename T, int RT, int CT>
__device__ __forceinline__ void tile_mma(f32x4 (&acc)[RT][CT], const char* ldsA, int rowA0, const char* ldsB, int rowB0) {
    using X = Tr<T>;
    const int lane = lane_id(), r16 = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        typename X::frag a[RT], b[CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = lds_frag<T>(ldsA, rowA0 + 16 * rt + r16, 4 * ks + lg);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[ct] = lds_frag<T>(ldsB, rowB0 + 16 * ct + r16, 4 * ks + lg);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = X::mma(a[rt], b[ct], acc[rt][ct]);
    }
}

}  // namespace aecf
