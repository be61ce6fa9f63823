// Shared helpers of the packed session kernels (seqp_fwd.hip, seqp_bwd.hip): addressing of tile-ordered and compact tensors.
// gfx950 only.
#pragma once
#include "seq_common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16_bf16(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

#define OOBH 0x40000000u            // out-of-range half: OOBH + OOBH stays out of range and does not wrap

// rows [first_row, first_row + nrows) of a [., row_elems] fp32 tensor
__device__ __forceinline__ Out make_rows(const void* base, size_t first_row, int nrows, int row_elems) {
    Out o;
    o.r = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)base + first_row * row_elems), 0, nrows * row_elems * 4, 0x00020000);
    o.sub = 0u;
    return o;
}
// byte offset of row t's record in a per-row tensor with rb bytes per row: the tile row itself, or -- pruned block -- the
// compact row b of a session's LAST position (other rows fall out of range)
__device__ __forceinline__ uint32_t row_base(bool pruned, int t, uint32_t rb, const int* info_l) {
    if (!pruned) return (uint32_t)t * rb;
    const int inf = info_l[t];
    return (inf & 64) ? (uint32_t)(inf >> 16) * rb : OOBH;
}

