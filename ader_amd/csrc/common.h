// Shared device helpers for the ader_amd HIP kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ADER_WAVE 64
#define NEG_PAD (-4294967296.0f)   // float32 value of the reference's -2**32+1 (modules.py:192,201)
#define LN_EPS 1e-8f               // modules.py:24

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Launch-side caches (hipFuncSetAttribute done, CU count, largest dynamic-LDS size granted) are kept PER DEVICE: a function attribute
// belongs to the device that was current when it was set, and a process may drive engines on several devices.
#define ADER_MAX_DEV 32
static inline int ader_cur_dev() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= ADER_MAX_DEV) d = 0;
    return d;
}

// status bits written by kernels into the engine's status word
#define ADER_ST_BAD_ID 1

// ---------------------------------------------------------------- counter-based dropout mask
// keep(idx) = (lowbias32(idx ^ key) >> 8) >= thr ; key = f(seed, step, site) computed on the host.
// The same spec is restated in oracle/ader_ref_cpu.py (dropout_keep) so masks can be compared.
struct DropArgs {
    uint32_t key;     // per (seed, step, site)
    uint32_t thr;     // round(rate * 2^24); 0 => dropout disabled
    float scale;      // float32(1)/(float32(1)-float32(rate))
    uint32_t base;    // counter offset of the local elements below `split`: global_row0 * elems_per_row (data-parallel shards)
    uint32_t split;   // first local element index of the SECOND row segment (exemplar rows follow the train rows, main.py:229;
    uint32_t base2;   // a shard holds a slice of each, so the two segments have different global offsets); 0xFFFFFFFF: one segment
};

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352DU; x ^= x >> 15; x *= 0x846CA68BU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool drop_keep(const DropArgs& d, uint32_t idx) {
    return (lowbias32((idx + (idx >= d.split ? d.base2 : d.base)) ^ d.key) >> 8) >= d.thr;
}
// host side of every launcher: the C-ABI descriptor (include/ader_hip.h: AderDrop; NULL = no dropout) -> kernel argument
template <class D> static inline DropArgs drop_from(const D* d) {
    DropArgs o;
    if (d) { o.key = d->key; o.thr = d->thr; o.scale = d->scale; o.base = d->base; o.split = d->split; o.base2 = d->base2; }
    else { o.key = 0; o.thr = 0; o.scale = 1.0f; o.base = 0; o.split = 0xFFFFFFFFu; o.base2 = 0; }
    return o;
}
// TF2 inverted dropout: (x * scale) * keep
__device__ __forceinline__ float drop_apply(const DropArgs& d, uint32_t idx, float x) {
    if (d.thr == 0) return x;
    return drop_keep(d, idx) ? x * d.scale : 0.0f;
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- f32 MFMA tile engine
// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16]; exact f32 fma chain in k order.
// Lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; D: col = l&15, row = (l>>4)*4 + reg.
// Operands are single floats per lane, so any LDS layout is expressed by (row stride, k stride).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[j] += A[m0:m0+16, 0:4*ksteps] * B[0:4*ksteps, n0+16j : n0+16j+16],  j = 0..NB-1
//   A element (i,k) at A[i*a_rs + k*a_ks], B element (k,n) at B[k*b_ks + n*b_ns]   (LDS or global)
template <int NB>
__device__ __forceinline__ void mma_tile(const float* __restrict__ A, int a_rs, int a_ks,
                                         const float* __restrict__ B, int b_ks, int b_ns,
                                         int ksteps, f32x4 (&acc)[NB], int lane) {
    const int r = lane & 15, q = lane >> 4;
    const float* ap = A + r * a_rs + q * a_ks;
    const float* bp = B + q * b_ks + r * b_ns;
    for (int ks = 0; ks < ksteps; ++ks) {
        const float a = ap[0];
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = mfma16(a, bp[j * 16 * b_ns], acc[j]);
        ap += 4 * a_ks;
        bp += 4 * b_ks;
    }
}

#define HIP_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
