// Tile image and transposed LDS reads shared by the batch-reduction kernels (aecf_gemm_tn_tr.hip, aecf_gemm_tn_hilo.hip).
//
// gfx950 reads an MFMA operand TRANSPOSED out of LDS (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block of
// 16-bit values and every lane receives one column), so a [batch row][feature] tile feeds a product whose reduction index is the
// batch.  Image of a [rows][128 bf16] tile: 8-row x 32-column subtiles of 512 B, 16-byte chunk ch of row r at
//     2048 (r >> 3) + 512 (ch >> 2) + 64 (r & 7) + 16 ((ch & 3) ^ ((r >> 2) & 3))
// (cdna_hip_programming.md T10, image (a)): LDS-DMA writes, ds_write_b128 and the transposed reads are bank-conflict free.
#pragma once
#include "aecf_common.h"

namespace aecf {

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;

__device__ __forceinline__ int tr_off(int row, int ch) {
    return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// MFMA operand (8 consecutive batch rows 32 ks + 8 lg .. + 7 of feature column col0 + r16) by two transposed reads
__device__ __forceinline__ u32x4 tr_frag(const char* tile, int addr_lo, int addr_hi) {
    const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(tile + addr_lo));
    const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(tile + addr_hi));
    const u32x2 l = __builtin_bit_cast(u32x2, lo), h = __builtin_bit_cast(u32x2, hi);
    return u32x4{l[0], l[1], h[0], h[1]};
}

}  // namespace aecf
