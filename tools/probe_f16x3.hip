// Probe for the "22-bit split" question of round 5's review (one experiment, not a kernel): what would fp16 hi/lo operand pieces buy over
// the bf16 hi/lo pieces of the x3 kernels (three MFMAs per product: hi.hi + hi.lo + lo.hi, fp32 accumulate)?
//   bf16x3   a = bf16(a) + bf16(a - hi):  8 + 8 mantissa bits per operand, 8-bit exponent (what k_lx3p / k_tab32x3 / gemm_x3 run)
//   f16x3u   a = f16(a) + f16(a - hi):    11 + 11 bits, but the lo piece of anything below 2^-3 is a SUBNORMAL fp16 (or flushed)
//   f16x3p   the same on operands PRE-SCALED by exact powers of two (table x 2^8, representations x 2^3), result scaled back
//   f16x3s   lo pieces scaled by 2^11 into their own accumulator: acc1 += hi.hi, acc2 += hi.lo' + lo'.hi, c = acc1 + 2^-11 acc2
//   f32      sequential fp32 fma chain (what "float32 matmul", ADER.py:91-93, means on a CPU)
// against an fp64 reference, on tiles shaped like the logit GEMM (32 representation rows x 32 items, K = 160) with the value ranges of
// (a) a freshly initialised 1M-item table (Glorot +-0.00245), (b) a trained table (N(0, 0.05)), (c) small representations.
// Error measure per element: |c - c_ref| / sum_k |a_k b_k| (the product-relative error the x3 bound 2^-16 is stated in).
// Also: does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs on gfx950?
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_f16x3 tools/probe_f16x3.hip && /tmp/probe_f16x3
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define KK 160
#define TILES 512

__device__ __forceinline__ float bf16_round(float x) {      // round to nearest even to bf16, returned as float
    uint32_t u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}
__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }

// one wave per tile; A [tile][32][KK], B [tile][KK][32], C [tile][32][32]
template <int MODE>
__global__ __launch_bounds__(64) void k_probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, float sa, float sb) {
    const int t = blockIdx.x, l = threadIdx.x, i = l & 31, kb = l >> 5;
    const float* a = A + (size_t)t * 32 * KK + (size_t)i * KK;
    const float* b = B + (size_t)t * KK * 32 + i;            // column j = l & 31
    f32x16 acc1 = {0}, acc2 = {0};
    if (MODE == 4) {                                          // fp32 fma chain, lane (i, kb) computes 16 of the 32 columns
        for (int r = 0; r < 16; ++r) {
            const int j = r * 2 + kb;
            float s = 0.0f;
            for (int k = 0; k < KK; ++k) s = fmaf(a[k], B[(size_t)t * KK * 32 + (size_t)k * 32 + j], s);
            C[(size_t)t * 1024 + i * 32 + j] = s;
        }
        return;
    }
    for (int k0 = 0; k0 < KK; k0 += 16) {
        float av[8], bv[8];
        for (int u = 0; u < 8; ++u) {
            av[u] = a[k0 + 8 * kb + u] * sa;
            bv[u] = b[(size_t)(k0 + 8 * kb + u) * 32] * sb;
        }
        if (MODE == 0) {
            bf16x8 ah, al, bh, bl;
            for (int u = 0; u < 8; ++u) {
                const float h = bf16_round(av[u]), g = bf16_round(bv[u]);
                ah[u] = to_bf16(h); al[u] = to_bf16(av[u] - h); bh[u] = to_bf16(g); bl[u] = to_bf16(bv[u] - g);
            }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc1, 0, 0, 0);
        } else {
            half8 ah, al, bh, bl;
            const float ls = (MODE == 3) ? 2048.0f : 1.0f;
            for (int u = 0; u < 8; ++u) {
                const _Float16 h = (_Float16)av[u], g = (_Float16)bv[u];
                ah[u] = h; al[u] = (_Float16)((av[u] - (float)h) * ls); bh[u] = g; bl[u] = (_Float16)((bv[u] - (float)g) * ls);
            }
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
            if (MODE == 3) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
            } else {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
            }
        }
    }
    const float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r >> 2) + 4 * kb + (r & 3);
        float v = acc1[r];
        if (MODE == 3) v = fmaf(acc2[r], 1.0f / 2048.0f, v);
        C[(size_t)t * 1024 + row * 32 + i] = v * inv;
    }
}

__global__ void k_subnormal(float* out) {
    half8 a, b;
    f32x16 c = {0};
    for (int u = 0; u < 8; ++u) { a[u] = (_Float16)0.0f; b[u] = (_Float16)0.0f; }
    // lane 0 (row 0 / col 0, k = 0): a = 2^-20 (fp16 subnormal: min normal 2^-14), b = 2^10 -> 2^-10 if subnormal inputs are kept
    if (threadIdx.x == 0) { a[0] = (_Float16)9.5367431640625e-07f; b[0] = (_Float16)1024.0f; }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

static double urand() { return (rand() + 0.5) / ((double)RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
    const size_t na = (size_t)TILES * 32 * KK, nb = (size_t)TILES * KK * 32, nc = (size_t)TILES * 1024;
    std::vector<float> A(na), B(nb), C(nc);
    float *dA, *dB, *dC, *dS;
    hipMalloc(&dA, na * 4); hipMalloc(&dB, nb * 4); hipMalloc(&dC, nc * 4); hipMalloc(&dS, 4);
    hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, dS);
    float sub = -1.0f;
    hipMemcpy(&sub, dS, 4, hipMemcpyDeviceToHost);
    printf("v_mfma_f32_32x32x16_f16 on an fp16 SUBNORMAL input (2^-20 x 2^10): %.10g  (kept: 0.0009765625, flushed: 0)\n", sub);
    const char* cases[3] = {"init table: rep ~ N(0,1), E ~ U(+-0.00245)", "trained table: rep ~ N(0,1), E ~ N(0,0.05)", "small rep: rep ~ N(0,0.01), E ~ N(0,0.05)"};
    const char* modes[5] = {"bf16x3", "f16x3u (unscaled lo)", "f16x3p (operands x 2^3 / x 2^8)", "f16x3s (lo x 2^11, 2 accumulators)", "f32 fma chain"};
    for (int cs = 0; cs < 3; ++cs) {
        srand(1234 + cs);
        for (size_t i = 0; i < na; ++i) A[i] = (float)(nrand() * (cs == 2 ? 0.01 : 1.0));
        for (size_t i = 0; i < nb; ++i) B[i] = (float)(cs == 0 ? (urand() * 2 - 1) * 0.00245 : nrand() * 0.05);
        hipMemcpy(dA, A.data(), na * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), nb * 4, hipMemcpyHostToDevice);
        std::vector<double> ref(nc), den(nc);
        for (int t = 0; t < TILES; ++t)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0, d = 0;
                    for (int k = 0; k < KK; ++k) {
                        const double p = (double)A[(size_t)t * 32 * KK + (size_t)i * KK + k] * (double)B[(size_t)t * KK * 32 + (size_t)k * 32 + j];
                        s += p; d += fabs(p);
                    }
                    ref[(size_t)t * 1024 + i * 32 + j] = s; den[(size_t)t * 1024 + i * 32 + j] = d;
                }
        printf("== %s\n", cases[cs]);
        for (int m = 0; m < 5; ++m) {
            const float sa = (m == 2) ? 8.0f : 1.0f, sb = (m == 2) ? 256.0f : 1.0f;
            hipMemset(dC, 0, nc * 4);
            if (m == 0) hipLaunchKernelGGL(k_probe<0>, dim3(TILES), dim3(64), 0, 0, dA, dB, dC, sa, sb);
            if (m == 1) hipLaunchKernelGGL(k_probe<1>, dim3(TILES), dim3(64), 0, 0, dA, dB, dC, sa, sb);
            if (m == 2) hipLaunchKernelGGL(k_probe<2>, dim3(TILES), dim3(64), 0, 0, dA, dB, dC, sa, sb);
            if (m == 3) hipLaunchKernelGGL(k_probe<3>, dim3(TILES), dim3(64), 0, 0, dA, dB, dC, sa, sb);
            if (m == 4) hipLaunchKernelGGL(k_probe<4>, dim3(TILES), dim3(64), 0, 0, dA, dB, dC, sa, sb);
            hipMemcpy(C.data(), dC, nc * 4, hipMemcpyDeviceToHost);
            double mx = 0, ss = 0;
            for (size_t i = 0; i < nc; ++i) {
                const double e = fabs((double)C[i] - ref[i]) / den[i];
                if (e > mx) mx = e;
                ss += e * e;
            }
            printf("   %-36s max %.3e (2^%.1f)   rms %.3e (2^%.1f)\n", modes[m], mx, log2(mx), sqrt(ss / nc), log2(sqrt(ss / nc)));
        }
    }
    return 0;
}
