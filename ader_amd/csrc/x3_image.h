// LDS image of a [32 rows][160 channels] bf16 operand tile, hi and lo planes, for the 16x16x32 x3 kernels
// (table_update_x3.hip: rows = batch rows of a rep chunk; logits_x3.hip: rows = items of a table block).  gfx950 only.
#pragma once
#include "lbf_common.h"

// One plane: 20 k-chunks (8 channels = 16 B per row) of [32 rows][16 B] = 512 B each, in quads of four:
//   byte offset of k-chunk kc = 2176 (kc >> 2) + 1152 ((kc >> 1) & 1) + 512 (kc & 1)
// k-chunks kc, kc+1 (kc even) are 512 B apart: the ds_read_b128 row read of lane (row c16, k-group g) at k-chunk 4 ks + g puts the
// 16 lanes of a read group (all 16 rows, two adjacent g) on 16 different 16-byte slots; k-chunks kc, kc+2 are 1152 B = 128 (mod 256)
// apart: the transposed read of a 4-row x 16-channel block built from k-chunks (kc, kc+2) covers all 64 banks once per 32 lanes.
#define X3_QUAD 2176
#define X3_PLANE_B (5 * X3_QUAD)   // bytes of one plane image (10,880)
#define X3_IMG_B 22528             // bytes of a chunk image in memory and in LDS: hi plane, lo plane, zero padding to 22 KiB
#define X3_BUF (X3_IMG_B / 2)      // bf16 elements of one LDS buffer
#define X3_PIECES (X3_IMG_B / 1024)
__host__ __device__ __forceinline__ constexpr int x3_kc_off(int kc) { return X3_QUAD * (kc >> 2) + 1152 * ((kc >> 1) & 1) + 512 * (kc & 1); }

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16_bf16(bf16x8 a, bf16x8 b, f32x4v c) {
#ifdef ADER_X3_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
// channel of lane c16 in the 16-channel block cb of a transposed read / of the product it feeds (k-chunks 4Q + o and 4Q + o + 2
// with Q = cb >> 1, o = cb & 1)
__device__ __forceinline__ int x3_channel(int cb, int c16) { return 32 * (cb >> 1) + 8 * (cb & 1) + 16 * (c16 >> 3) + (c16 & 7); }

// ---- the same tile for the 32x32x16 operand maps (logits_x3.hip: k_lx3g): k-chunk kc at byte offset 576 kc (512 + 64 of padding).
// Row read of lane (row r = lane & 31, k-half hh) at k-chunk 2 ks + hh: a ds_read_b128 group holds 16 different rows of ONE
// k-chunk = 16 different 16-byte slots.  Transposed read of a 4-row x 16-channel block: the 32 lanes of a half read 4 consecutive
// rows (64 B) of 4 consecutive k-chunks, which sit 576 = 64 (mod 256) bytes apart: all 64 banks once.
#define X3B_KC 576
#define X3B_PLANE_B (20 * X3B_KC)  // 11,520
#define X3B_IMG_B (2 * X3B_PLANE_B)
