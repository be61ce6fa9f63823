// Shared device helpers of the session-tiled kernels (seq_fwd.hip, seq_bwd.hip): one workgroup of 10 waves owns the T <= 64
// rows of a session; GEMM operands are bf16 hi/lo tile pairs [64][168] in LDS, the fp32 working tile is Xf [64][164].
// gfx950 only.
#pragma once
#include "common.h"
#include "../../include/ader_hip.h"
#include <stddef.h>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define HP 160
#define LDR 168
#define WSZ (HP * LDR)
#define TR 64
#define XS 164                      // fp32 row stride of Xf
#define RSZ (2 * TR * LDR)          // bf16 elements of one hi/lo tile pair
#define LDP 72                      // row stride of the 64x64 probability tile

static_assert(sizeof(AderDrop) == sizeof(DropArgs), "AderDrop must mirror DropArgs");
static_assert(TR * XS * sizeof(float) <= RSZ * sizeof(bf16), "Xf must fit in one tile pair");

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }
__device__ __forceinline__ bf16x4 tr_read(const bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
    return o;
}
// Launders a lane-derived index so that the offsets computed from it are rebuilt per phase instead of being hoisted out of
// the block loop and kept (spilled) across all phases.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
#define PHASE_IDS                                             \
    const int lane_p = opaque(lane);                          \
    const int r = lane_p & 31, hh = lane_p >> 5;              \
    const int n = 32 * nb + r;                                \
    (void)hh; (void)n
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store of the wave
// (vmcnt(0)): with ~100 activation stores per lane between barriers that costs a memory round trip per phase.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <class D> __device__ __forceinline__ DropArgs drop_of(const D& d) {
    DropArgs o; o.key = d.key; o.thr = d.thr; o.scale = d.scale; o.base = d.base; o.split = d.split; o.base2 = d.base2;
    return o;
}
// Dropout of the session-tiled kernels: all elements of a workgroup lie in ONE row segment of the counter space (the segments
// split at whole sessions), so the segment base is chosen once per phase and folded into the lane's base index, and the keep
// decision is branch-free -- thr == 0 (dropout off) keeps everything at scale 1.  (The per-element "thr == 0?" branch of
// drop_apply cut the 16-element epilogues into 16 basic blocks whose dependent chains -- two quarter-rate integer multiplies
// each -- could not be interleaved: a wave took ~4 us per epilogue with the SIMD mostly idle.)  Same counters, same decisions.
struct SDrop { uint32_t key, thr, off; float scale; };
template <class D> __device__ __forceinline__ SDrop sdrop_of(const D& d, uint32_t first_idx) {
    SDrop o;
    o.key = d.key; o.thr = d.thr; o.scale = d.thr ? d.scale : 1.0f;
    o.off = (first_idx >= d.split) ? d.base2 : d.base;
    return o;
}
// idx_off = local element index + SDrop.off.  ON is the workgroup-uniform "dropout enabled" decision, taken ONCE per phase by
// the caller (two copies of the unrolled loop): evaluation, herding and finetune runs must not pay for the hash.
template <bool ON> __device__ __forceinline__ float sdrop_apply(const SDrop& d, uint32_t idx_off, float x) {
    if (!ON) return x;
    return ((lowbias32(idx_off ^ d.key) >> 8) >= d.thr) ? x * d.scale : 0.0f;
}
// single elements (not worth a second code path)
__device__ __forceinline__ float sdrop_apply1(const SDrop& d, uint32_t idx_off, float x) {
    return d.thr ? sdrop_apply<true>(d, idx_off, x) : x;
}
__device__ __forceinline__ void put_split(bf16* Th, bf16* Tl, int off, float v) {
    const bf16 h = (bf16)v;
    Th[off] = h;
    Tl[off] = (bf16)(v - (float)h);
}

// B fragments of this wave's 32 output columns: planes in fragment order (k_wprep: the 64 lanes' operands of a fragment are 1 KiB
// contiguous; W^T hi at W, lo at W + WSZ), zero padded
__device__ __forceinline__ void load_bfrags(const bf16* __restrict__ W, int nb, int r, int hh, bf16x8 (&bh)[10], bf16x8 (&bl)[10]) {
    const bf16* p = W + ((size_t)nb * 10 * 64 + 32 * hh + r) * 8;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        bh[ks] = *(const bf16x8*)(p + 512 * ks);
        bl[ks] = *(const bf16x8*)(p + WSZ + 512 * ks);
    }
}
// acc = tile rows 32mh.. (hi/lo in LDS) . W columns 32nb..
// skip (wave-uniform): the wave's 32 rows are all leading padding -- nobody consumes the product (on the shipped data, mean session
// length 5 of 50 positions, that is half of the waves of most workgroups: their LDS reads and MFMAs only slowed the other half down)
__device__ __forceinline__ f32x16 tile_mma(const bf16* Th, int mh, int r, int hh, const bf16x8 (&bh)[10], const bf16x8 (&bl)[10],
                                           bool skip = false) {
    const bf16* Ah = Th + (32 * mh + r) * LDR + 8 * hh;
    const bf16* Al = Ah + TR * LDR;
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    if (skip) return acc;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        const bf16x8 ah = *(const bf16x8*)(Ah + 16 * ks);
        const bf16x8 al = *(const bf16x8*)(Al + 16 * ks);
        acc = mfma_bf16(al, bh[ks], acc);
        acc = mfma_bf16(ah, bl[ks], acc);
        acc = mfma_bf16(ah, bh[ks], acc);
    }
    return acc;
}

// LayerNorm of one fp32 LDS row (modules.py:44-48), one wave; x[i], y[i], gamma g[i], beta be[i] for columns lane + 64 i
__device__ __forceinline__ void ln_row(const float* xr, bool valid, int H, int lane, const float (&g)[3], const float (&be)[3],
                                       float (&x)[3], float (&y)[3], float& mean, float& sd, float& xsum, float& ysum) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        x[i] = (valid && c < H) ? xr[c] : 0.0f;
        s += x[i];
    }
    s = wave_sum(s);
    mean = s / (float)H;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        const float dlt = (c < H) ? (x[i] - mean) : 0.0f;
        q += dlt * dlt;
    }
    q = wave_sum(q);
    sd = sqrtf(q / (float)H + LN_EPS);
    float ys = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        y[i] = 0.0f;
        if (valid && c < H) {
            y[i] = g[i] * ((x[i] - mean) / sd) + be[i];   // a true division: bit-equal to k_ln_fwd
            ys += y[i];
        }
    }
    xsum = s;
    ysum = wave_sum(ys);
}
// Sum over the 16 lanes of a DPP row, result in every lane: xor-1 and xor-2 inside quads, then the two mirror swaps.
// (4 VALU ops with DPP modifiers; a __shfl_xor is an LDS round trip.)
#define DPP_F(v_, ctrl_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v_)), (ctrl_), 0xf, 0xf, false))
__device__ __forceinline__ float row16_sum(float v) {
    v += DPP_F(v, 0xB1);      // quad_perm [1,0,3,2]
    v += DPP_F(v, 0x4E);      // quad_perm [2,3,0,1]
    v += DPP_F(v, 0x141);     // row_half_mirror
    v += DPP_F(v, 0x140);     // row_mirror
    return v;
}
// per-column parameters in the 16-lanes-per-row layout: column sub + 16 i
__device__ __forceinline__ void load10(const float* __restrict__ p, int H, int sub, float (&o)[10]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int c = sub + 16 * i;
        o[i] = (c < H) ? p[c] : 0.0f;
    }
}
__device__ __forceinline__ void load3(const float* __restrict__ p, int H, int lane, float (&o)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        o[i] = (c < H) ? p[c] : 0.0f;
    }
}

// ---- bounds-checked activation traffic -------------------------------------------------------------------------------
// Every activation store/load of a phase goes through a raw buffer descriptor that covers exactly the rows of THIS session
// (or, for a pruned block, the single compact row of position T-1): rows >= T, columns >= H and the non-kept rows fall
// outside the descriptor and are dropped by the hardware range check instead of by per-element branches, and the address
// is one 32-bit add per element.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define OOB 0x80000000u
struct Out { rsrc_t r; uint32_t sub; };
// rows_elems: elements per row of the tensor (H, 1 or T); the tensor is [B*T][rows_elems] or, pruned, [B][rows_elems]
__device__ __forceinline__ Out make_out(const void* base, int b, int T, int row_elems, bool pruned) {
    Out o;
    const size_t first = pruned ? (size_t)b * row_elems : (size_t)b * T * row_elems;
    o.r = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)base + first), 0, (pruned ? row_elems : T * row_elems) * 4, 0x00020000);
    o.sub = pruned ? (uint32_t)(T - 1) * row_elems * 4u : 0u;
    return o;
}
__device__ __forceinline__ void bstore(const Out& o, uint32_t boff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), o.r, (int)(boff - o.sub), 0, 2);
}
__device__ __forceinline__ float bload(const Out& o, uint32_t boff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(o.r, (int)(boff - o.sub), 0, 0));
}
#define ROWJ(j) (((j) & 3) + 8 * ((j) >> 2))          // row of accumulator register j relative to 32mh + 4hh
// In a pruned (last) block only row T-1 of the query / FFN path is consumed (rows are independent there; K and V still need every
// row): the waves that do not own that row skip those products (act), and the owning waves run the epilogue for the ONE
// accumulator register that holds it -- row T-1 = 32 mhT + 4 hhT + ROWJ(jT).  The other rows of the tiles keep stale contents
// that nothing consumes.  Rebuilt per phase from a laundered T (scalar registers are scarce in these kernels).
__device__ __forceinline__ int opaque_s(int v) { asm volatile("" : "+s"(v)); return v; }
#define PRUNE_IDS                                                                      \
    const int rT = opaque_s(T) - 1;                                                    \
    const int mhT = rT >> 5, hhT = ((rT & 31) >> 2) & 1;                               \
    const int jT = 4 * ((rT & 31) >> 3) + (rT & 3), rjT = (rT & 3) + 8 * ((rT & 31) >> 3); \
    const bool act = !pruned || mh == mhT;                                             \
    (void)hhT; (void)jT; (void)rjT
// accumulator register j of a tile (j wave-uniform, known only at run time): a chain of 15 selects on a scalar condition
__device__ __forceinline__ float pick16(const f32x16& a, int j) {
    float v = a[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) v = (j == i) ? a[i] : v;
    return v;
}

