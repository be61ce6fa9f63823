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


// ---- small-tile mapping: wave w owns output columns 16 w .. 16 w + 15 of the tile's (at most 32) rows.  v_mfma_f32_16x16x32_bf16: lane (c = lane & 15,
// g = lane >> 4) supplies A[row c][k = 8 g ..] and B[k = 8 g ..][column c] and holds D[rows 4 g + i][column c].
// B fragments from the planes of k_wprep (fragment order of the 32x32x16 maps): the 16 bytes (n, k .. k + 7) a lane needs are one chunk there
__device__ __forceinline__ void load_bfrags16(const bf16* __restrict__ W, int w, int lane, bf16x8 (&bh)[10], bf16x8 (&bl)[10]) {
    const int c = lane & 15, g = lane >> 4;
    const bf16* p = W + ((size_t)(w >> 1) * 10 * 64 + 32 * (g & 1) + 16 * (w & 1) + c) * 8 + (g >> 1) * 512;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        bh[ks] = *(const bf16x8*)(p + 1024 * ks);
        bl[ks] = *(const bf16x8*)(p + WSZ + 1024 * ks);
    }
}
// acc[rb] = tile rows 16 rb .. (hi/lo in LDS) . W columns 16 w ..
__device__ __forceinline__ void tile_mma16(const bf16* Th, int c, int g, const bf16x8 (&bh)[10], const bf16x8 (&bl)[10], int nrb, f32x4v (&acc)[2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        acc[rb] = (f32x4v){0.0f, 0.0f, 0.0f, 0.0f};
        if (rb >= nrb) continue;
        const bf16* Ah = Th + (16 * rb + c) * LDR + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const bf16x8 ah = *(const bf16x8*)(Ah + 32 * ks);
            const bf16x8 al = *(const bf16x8*)(Ah + TR * LDR + 32 * ks);
            acc[rb] = mfma16_bf16(al, bh[ks], acc[rb]);
            acc[rb] = mfma16_bf16(ah, bl[ks], acc[rb]);
            acc[rb] = mfma16_bf16(ah, bh[ks], acc[rb]);
        }
    }
}

