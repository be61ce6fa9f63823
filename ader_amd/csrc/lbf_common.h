// Shared pieces of the bf16-MFMA logit kernels (logits_bf16.hip, logits_x3.hip, table_update.hip).  gfx950 only.
#pragma once
#include "common.h"

// -DADER_X3_F16 (experiment of round 6, tools/ab_f16x3.sh; NOT the product build): the operand pieces of the float32-grade logit
// FORWARD (k_lx3_prep, k_lx3p) as fp16 hi / lo on operands pre-scaled by exact powers of two -- 11 + 11 mantissa bits instead of 8 + 8 at
// the same three MFMAs (tools/probe_f16x3.hip).  The scales keep the lo pieces in fp16's normal range: representations x 2^3, table
// x 2^8, probabilities x 2^8 (added to the exponent); the accumulators are scaled back where the kernel stores them.  Only the forward
// entry points are valid in such a build (the update kernels still read the planes as bf16).
#ifdef ADER_X3_F16
typedef _Float16 bf16;
#define X3_SR 8.0f
#define X3_SE 256.0f
#define X3_SPL 8.0f
#else
typedef __bf16 bf16;
#define X3_SR 1.0f
#define X3_SE 1.0f
#define X3_SPL 0.0f
#endif
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define HP 160
#define LDR 168                 // bf16 elements per LDS / rep_bf row
#define LOG2E 1.4426950408889634f
#define RESCALE_THR 6.0f        // lazy online-softmax rescale threshold (log2 units): p <= 2^6

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef ADER_X3_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
// accumulator row of register `reg` for lane half hh (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// 4(k) x 16(n) transposed LDS read: lane (q = (lane&15)>>2, p = lane&3) supplies the address of row k0+q, cols n0+4p..;
// lane i of the 16-lane group receives column n0+i of rows k0..k0+3.
#ifdef ADER_X3_F16
typedef short i16x4_tr __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 tr_read(const bf16* p) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4_tr __attribute__((address_space(3)))*)p));
}
#else
__device__ __forceinline__ bf16x4 tr_read(const bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p);
}
#endif

typedef __attribute__((address_space(3))) bf16 lds_bf16;
__device__ __forceinline__ bf16x4 tr_read3(const lds_bf16* p) {
#ifdef ADER_X3_F16
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4_tr __attribute__((address_space(3)))*)p));
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p);
#endif
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)v[8 * s + j];
    return o;
}


// workgroup barrier that orders LDS traffic only (__syncthreads also drains every outstanding global access of the wave)
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int lower_bound_i32(const int* __restrict__ a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// Sparse terms and optimiser constants of the fused table update (table_update.hip): when the whole gradient of a table row is
// available inside the workgroup that owns it -- the dense logits term from the MFMAs, plus the sparse input-embedding rows and
// one-hot target rows looked up in id-sorted lists -- the TF-Adam update of that row (ADER.py:96) is applied in place.
struct FuseArgs {
    const int* sp_ids; const int* sp_rows; int n_sp; const float* sp_src; float sp_scale;   // input-embedding rows (sorted by id)
    const int* tg_ids; const int* tg_rows; int n_tg; const float* wrow;                      // one-hot targets (sorted by id)
    const int* tile_meta;       // per 64-row tile: [2 lists][18] = {k0, k1, first 8 (id, row) entries} (ader_tab_tile_meta)
    float* emb1; float* m1; float* v1; bf16* sh1w;   // row of item 1 of theta/m/v and of the bf16 shadow (NULL: no shadow)
    float lr_t, omb1, omb2, eps;
    const float* extra1;        // EXTRA: dense gradient rows to add (row of item 1; [.,H] fp32), e.g. distilled rows' term
};

// arguments of the 64-row table-update kernels (table_update.hip: k_tab_upd, table_update_x3.hip: k_tab16x3)
struct TabArgs {
    const float* emb1;      // fp32 table, row of item 1: GEMM operand source (and the parameters, FuseArgs.emb1 == this)
    int vrows;              // table rows that exist from emb1 on (item_num)
    const bf16* rep_hi;     // [Bp][LDR] bf16(rep), zero padded
    const bf16* rep_lo;     // [Bp][LDR] bf16(rep - hi)   (X3)
    const void* rep_img;    // k_tab16x3: LDS images of the rep chunks (ader_x3_rep_image); NULL elsewhere
    int ko;                 // timing-only knock-outs (ADER_X3_KO; 0 in every real run): 1 no theta/m/v traffic, 2 no GEMM, 4 no DMA, 8 no barrier
    const float* off;       // [Bp] log2(w_b) - lse2_b; -inf for rows without a loss term
    int Bp, H, N, tile_off;
    int tile_end;           // k_tab32x3: first 64-row tile beyond the launch (its workgroups own PAIRS of tiles)
    float* demb1;           // !ADAM: gradient row of item 1
    // KD rows (ADER.py:132-137): batch rows [kd_row0, Bp) are distilled exemplar rows: dlogit = w (softmax(s[:Np]) - softmax(t)),
    // zero for items >= Np.  kd_row0 % 128 == 0; = Bp: none.  trow / tlse2: [Bp] as written by ader_lx3_fwd_kd.
    int kd_row0, Np;
    const float* teacher; long ldt; const int* trow; const float* tlse2;
};

// arguments of the x3 flash forward kernels (logits_bf16.hip: k_lx3_fwd, logits_x3.hip: k_lx3g / k_lx3p / k_lx3r)
struct Lx3Args {
    const float* emb1;          // fp32 table, row of item 1
    int vrows;                  // table rows available from emb1 (item_num)
    const bf16* rep_hi; const bf16* rep_lo;     // [Bp][LDR]
    int Bp, H, N, ranges;
    float* pm; float* pl; float* pO;
    // distilled rows (as LbfArgs): rows [kd_row0, Bp) take the softmax over the first Np items and have a teacher readout chunk
    int kd_row0, Np;
    const float* teacher; long ldt; const int* trow; const float* tlse2; float* pO2;
    int ranges2;                // item ranges of the readout launch (it is a launch of its own in x3 mode: its own partition)
};
