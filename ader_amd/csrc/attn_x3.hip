// Causal masked self-attention core on the bf16 matrix cores with hi/lo operand splitting ("bf16x3", see gemm_x3.hip):
// float32-grade accuracy at a fraction of the f32-MFMA time.  Same semantics as attn.hip (reference multihead_attention,
// modules.py:177-223).  One workgroup (2 waves) per (sequence, head); wave w owns query rows 32w..32w+31.
//
// Forward : S^T = K.Q^T with the keys on the MFMA rows and the queries on the lanes, so the softmax statistics of a query
//           are lane-local; the probabilities go back into the matrix core straight from the accumulator as the A operand
//           of O += P^T . V, V read k-major with ds_read_b64_tr_b16.  Saves P^T [key][query] (softmax before the query mask).
// Backward: dP^T = V.dO^T, softmax backward lane-local, dQ from the accumulator operand (dS^T . K), dV = P_drop^T . dO and
//           dK = dS^T . Q through small LDS tiles.  One 64x168 hi/lo LDS tile pair is restaged per phase (K, V, dO, Q).
// T <= 64, dh = H/heads even and <= 160.  gfx950 only.
#include "common.h"
#include "../../include/ader_hip.h"

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define LDR 168
#define LDP 72                  // row stride (elements) of the 64x64 probability tiles
#define TR 64

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

struct AttnX3Args {
    const float* Q; const float* K; const float* V;   // [B,T,H]
    const float* res;                                   // fwd: LN'd queries (residual); bwd: dO
    const float* kmask; const float* qmask;            // [B,T]
    float* out;                                         // fwd: x1 [B,T,H]
    float* PT;                                          // [B,heads,T(key),T(query)]
    float* dQ; float* dK; float* dV;
    int B, T, H, heads;
    float sqrt_dh;
    DropArgs drop;
};

// dst_hi/lo[t][c] = split(src[t][c0 + c]) for t < T, c < dh (pairs); padding untouched (zeroed once by the caller)
template <int NT = 128>
__device__ __forceinline__ void stage_split(bf16* Th, bf16* Tl, const float* __restrict__ src, int T, int H, int c0, int dh, int tid) {
    const int HH = dh >> 1, n2 = T * HH;
    // 8 independent loads in flight per thread before any conversion (the tile comes from L2 / Infinity Cache: latency-bound)
    for (int i0 = 0; i0 < n2; i0 += NT * 8) {
        float2 v[8];
        int t[8], c2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i0 + tid + NT * u;
            t[u] = idx / HH; c2[u] = idx - t[u] * HH;
            v[u] = (idx < n2) ? *(const float2*)(src + (size_t)t[u] * H + c0 + 2 * c2[u]) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + tid + NT * u < n2) {
                bf16x2 h, l;
                h[0] = (bf16)v[u].x; h[1] = (bf16)v[u].y;
                l[0] = (bf16)(v[u].x - (float)h[0]); l[1] = (bf16)(v[u].y - (float)h[1]);
                *(bf16x2*)(Th + t[u] * LDR + 2 * c2[u]) = h;
                *(bf16x2*)(Tl + t[u] * LDR + 2 * c2[u]) = l;
            }
        }
    }
}

// operand fragments of one row of a [T,H] tensor: lane (row q, half hh) holds x[q][c0 + 16ks + 8hh + 0..7], split hi/lo
__device__ __forceinline__ void row_frags(const float* __restrict__ src, bool valid, int dh, int hh, bf16x8 (&fh)[10], bf16x8 (&fl)[10]) {
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            const int k = 16 * ks + 8 * hh + 2 * j2;
            float2 v = make_float2(0.f, 0.f);
            if (valid && k < dh) v = *(const float2*)(src + k);
            const bf16 h0 = (bf16)v.x, h1 = (bf16)v.y;
            fh[ks][2 * j2] = h0; fh[ks][2 * j2 + 1] = h1;
            fl[ks][2 * j2] = (bf16)(v.x - (float)h0); fl[ks][2 * j2 + 1] = (bf16)(v.y - (float)h1);
        }
    }
}

// acc[kb] (keys 32kb.. on rows, this wave's queries on lanes) = tile rows . frags^T   (3 MFMAs per product)
__device__ __forceinline__ void rows_times_frags(const bf16* Th, const bf16* Tl, const bf16x8 (&fh)[10], const bf16x8 (&fl)[10],
                                                 int ksteps, int r, int hh, f32x16 (&acc)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[kb][j] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            if (ks < ksteps) {
                const bf16x8 ah = *(const bf16x8*)(Th + (32 * kb + r) * LDR + 16 * ks + 8 * hh);
                const bf16x8 al = *(const bf16x8*)(Tl + (32 * kb + r) * LDR + 16 * ks + 8 * hh);
                acc[kb] = mfma_bf16(al, fh[ks], acc[kb]);
                acc[kb] = mfma_bf16(ah, fl[ks], acc[kb]);
                acc[kb] = mfma_bf16(ah, fh[ks], acc[kb]);
            }
        }
    }
}

// O[nb] += X^T . tile  where X = the two 32x32 accumulators x[kb] (rows = tile rows 32kb.., cols = this wave's lanes),
// fed back as A operands (hi/lo split of the fp32 values); tile rows read k-major in the accumulator's row order.
__device__ __forceinline__ void acc_times_rows(const f32x16 (&x)[2], const bf16* Th, const bf16* Tl, int nblocks, int lane,
                                               f32x16 (&O)[5], int nb0 = 0) {
    const int hh = lane >> 5, q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = x[kb][8 * s + j];
                ph[j] = (bf16)v;
                pl[j] = (bf16)(v - (float)ph[j]);
            }
            const int ro = (32 * kb + 16 * s + 4 * hh + q4) * LDR + 16 * g1 + 4 * p4;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                if (nb >= nb0 && nb < nblocks) {
                    const bf16x8 bh = cat4(tr_read(Th + ro + 32 * nb), tr_read(Th + ro + 32 * nb + 8 * LDR));
                    const bf16x8 bl = cat4(tr_read(Tl + ro + 32 * nb), tr_read(Tl + ro + 32 * nb + 8 * LDR));
                    O[nb] = mfma_bf16(pl, bh, O[nb]);
                    O[nb] = mfma_bf16(ph, bl, O[nb]);
                    O[nb] = mfma_bf16(ph, bh, O[nb]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(128) void k_attn_x3_fwd(AttnX3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* Th = (bf16*)smem_raw;
    bf16* Tl = Th + TR * LDR;
    float* km_l = (float*)(Tl + TR * LDR);                  // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x / a.heads, head = blockIdx.x % a.heads;
    const int T = a.T, H = a.H, dh = H / a.heads, c0 = head * dh;
    const size_t base = (size_t)b * T * H;
    const int ksteps = (dh + 15) >> 4, nblocks = (dh + 31) >> 5;
    const int q = 32 * wave + r;
    for (int i = tid; i < 2 * TR * LDR / 2; i += 128) ((uint32_t*)Th)[i] = 0u;
    if (tid < TR) km_l[tid] = (tid < T) ? a.kmask[(size_t)b * T + tid] : 0.0f;
    bf16x8 qh[10], ql[10];
    row_frags(a.Q + base + (size_t)(q < T ? q : 0) * H + c0, q < T, dh, hh, qh, ql);
    __syncthreads();
    stage_split(Th, Tl, a.K + base, T, H, c0, dh, tid);
    __syncthreads();
    f32x16 S[2];
    rows_times_frags(Th, Tl, qh, ql, ksteps, r, hh, S);
    // softmax over the T keys of query q (lane-local + the other half-wave)
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            float s = S[kb][j] / a.sqrt_dh;                              // modules.py:185
            if (key < T) {
                if (km_l[key] == 0.0f) s = NEG_PAD;                      // modules.py:188-193
                if (key > q) s = NEG_PAD;                                // modules.py:196-202
                mx = fmaxf(mx, s);
            }
            S[kb][j] = s;
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            const float e = (key < T) ? expf(S[kb][j] - mx) : 0.0f;
            S[kb][j] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float qm = (q < T) ? a.qmask[(size_t)b * T + q] : 0.0f;         // modules.py:208-211
    const size_t pbase = ((size_t)b * a.heads + head) * T;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            float p = 0.0f;
            if (key < T && q < T) {
                p = S[kb][j] / sum;
                a.PT[(pbase + key) * T + q] = p;
                p = drop_apply(a.drop, (uint32_t)((pbase + q) * T + key), p * qm);   // modules.py:214
            }
            S[kb][j] = p;
        }
    __syncthreads();                                        // all reads of the K tiles are done
    stage_split(Th, Tl, a.V + base, T, H, c0, dh, tid);
    __syncthreads();
    f32x16 O[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    acc_times_rows(S, Th, Tl, nblocks, lane, O);
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
        const int c = 32 * nb + r;
        if (c >= dh) continue;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int qq = 32 * wave + acc_row(j, hh);
            if (qq < T) {
                const size_t idx = base + (size_t)qq * H + c0 + c;
                a.out[idx] = O[nb][j] + a.res[idx];                      // modules.py:223
            }
        }
    }
}

// out[nb] = tileP rows (this wave's 32 rows, k = 64 columns) . tile rows (k-major read, natural k order), 3 MFMAs per product
__device__ __forceinline__ void ptile_times_rows(const bf16* Ph, const bf16* Pl, const bf16* Th, const bf16* Tl, int nblocks, int lane,
                                                 int wave, f32x16 (&O)[5], int nb0 = 0) {
    const int r = lane & 31, hh = lane >> 5, q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 ah = *(const bf16x8*)(Ph + (32 * wave + r) * LDP + 16 * ks + 8 * hh);
        const bf16x8 al = *(const bf16x8*)(Pl + (32 * wave + r) * LDP + 16 * ks + 8 * hh);
        const int ro = (16 * ks + 8 * hh + q4) * LDR + 16 * g1 + 4 * p4;
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            if (nb >= nb0 && nb < nblocks) {
                const bf16x8 bh = cat4(tr_read(Th + ro + 32 * nb), tr_read(Th + ro + 32 * nb + 4 * LDR));
                const bf16x8 bl = cat4(tr_read(Tl + ro + 32 * nb), tr_read(Tl + ro + 32 * nb + 4 * LDR));
                O[nb] = mfma_bf16(al, bh, O[nb]);
                O[nb] = mfma_bf16(ah, bl, O[nb]);
                O[nb] = mfma_bf16(ah, bh, O[nb]);
            }
        }
    }
}

__device__ __forceinline__ void store_rows(float* __restrict__ dst, const f32x16 (&O)[5], int wave, int lane, int T, int H, int c0, int dh,
                                           int nb0 = 0, int nb1 = 5) {
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
        const int c = 32 * nb + r;
        if (c >= dh || nb < nb0 || nb >= nb1) continue;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = 32 * wave + acc_row(j, hh);
            if (row < T) dst[(size_t)row * H + c0 + c] = O[nb][j];
        }
    }
}

// Four waves: wave (w2 = wave & 1, half = wave >> 1) owns query rows / key rows 32 w2.. as in the two-wave form, and of the three
// [T, dh] products (dQ, dV, dK) the 32-channel blocks {0, 1, 2} (half 0) or {3, 4} (half 1).  dP^T and the softmax backward of its
// 32 queries are computed by both halves (60 MFMAs and 32 probabilities: cheaper than an exchange); half 0 writes the P_drop / dS
// tiles.  With two waves the four operand stagings (V, K, dO, Q: memory -> registers -> LDS, each behind a barrier) ran on 128
// threads with one wave per SIMD on half of the CU's SIMDs: 62 us for 512 sessions at 6 % matrix-pipe use.
__global__ __launch_bounds__(256) void k_attn_x3_bwd(AttnX3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* Th = (bf16*)smem_raw;
    bf16* Tl = Th + TR * LDR;
    bf16* Pdh = Tl + TR * LDR;                              // P_drop^T [key][query] hi/lo, dS^T hi/lo : [64][LDP] each
    bf16* Pdl = Pdh + TR * LDP;
    bf16* dSh = Pdl + TR * LDP;
    bf16* dSl = dSh + TR * LDP;
    float* km_l = (float*)(dSl + TR * LDP);
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 1, half = tid >> 7;
    const int r = lane & 31, hh = lane >> 5;
    const int nb0 = half ? 3 : 0, nb1 = half ? 5 : 3;        // this wave's channel blocks of dQ / dV / dK
    const int b = blockIdx.x / a.heads, head = blockIdx.x % a.heads;
    const int T = a.T, H = a.H, dh = H / a.heads, c0 = head * dh;
    const size_t base = (size_t)b * T * H;
    const int ksteps = (dh + 15) >> 4, nblocks = (dh + 31) >> 5;
    const int q = 32 * wave + r;
    for (int i = tid; i < (2 * TR * LDR + 4 * TR * LDP) / 2; i += 256) ((uint32_t*)Th)[i] = 0u;
    if (tid < TR) km_l[tid] = (tid < T) ? a.kmask[(size_t)b * T + tid] : 0.0f;
    f32x16 X[2];                                            // dP^T, then dS^T (keys on rows, this wave's queries on lanes)
    {
        bf16x8 gh[10], gl[10];
        row_frags(a.res + base + (size_t)(q < T ? q : 0) * H + c0, q < T, dh, hh, gh, gl);      // dO rows
        __syncthreads();
        stage_split<256>(Th, Tl, a.V + base, T, H, c0, dh, tid);
        __syncthreads();
        rows_times_frags(Th, Tl, gh, gl, ksteps, r, hh, X);                                      // dP_drop^T = V . dO^T
    }
    // softmax backward for query q
    const float qm = (q < T) ? a.qmask[(size_t)b * T + q] : 0.0f;
    const size_t pbase = ((size_t)b * a.heads + head) * T;
    f32x16 Pv[2];
    float dot = 0.0f;
    // the 32 saved probabilities of this lane: UNCONDITIONAL loads (element 0 of the block where the entry does not exist) issued
    // together -- a load under the per-element branch below was waited for inside it: 32 serial memory round trips per wave
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            Pv[kb][j] = a.PT[(key < T && q < T) ? (pbase + key) * T + q : pbase * T];
        }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            float p = 0.0f, dp = 0.0f, pd = 0.0f;
            if (key < T && q < T) {
                p = Pv[kb][j];
                float f = qm;
                if (a.drop.thr != 0) f = drop_keep(a.drop, (uint32_t)((pbase + q) * T + key)) ? f * a.drop.scale : 0.0f;
                dp = X[kb][j] * f;
                pd = p * f;
                dot += dp * p;
            }
            Pv[kb][j] = p;
            X[kb][j] = dp;
            const bf16 ph = (bf16)pd;
            if (half == 0) {
                Pdh[key * LDP + q] = ph;
                Pdl[key * LDP + q] = (bf16)(pd - (float)ph);
            }
        }
    dot += __shfl_xor(dot, 32, 64);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            float ds = 0.0f;
            if (key < T && q < T && key <= q && km_l[key] != 0.0f) ds = (Pv[kb][j] * (X[kb][j] - dot)) / a.sqrt_dh;
            X[kb][j] = ds;
            const bf16 sh_ = (bf16)ds;
            if (half == 0) {
                dSh[key * LDP + q] = sh_;
                dSl[key * LDP + q] = (bf16)(ds - (float)sh_);
            }
        }
    f32x16 O[5];
    // dQ[q][c] = sum_key dS^T[key][q] K[key][c]   (accumulator operand)
    __syncthreads();
    stage_split<256>(Th, Tl, a.K + base, T, H, c0, dh, tid);
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    acc_times_rows(X, Th, Tl, min(nblocks, nb1), lane, O, nb0);
    store_rows(a.dQ + base, O, wave, lane, T, H, c0, dh, nb0, nb1);
    // dV[key][c] = sum_q P_drop^T[key][q] dO[q][c]   (wave = key block)
    __syncthreads();
    stage_split<256>(Th, Tl, a.res + base, T, H, c0, dh, tid);
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    ptile_times_rows(Pdh, Pdl, Th, Tl, min(nblocks, nb1), lane, wave, O, nb0);
    store_rows(a.dV + base, O, wave, lane, T, H, c0, dh, nb0, nb1);
    // dK[key][c] = sum_q dS^T[key][q] Q[q][c]
    __syncthreads();
    stage_split<256>(Th, Tl, a.Q + base, T, H, c0, dh, tid);
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    ptile_times_rows(dSh, dSl, Th, Tl, min(nblocks, nb1), lane, wave, O, nb0);
    store_rows(a.dK + base, O, wave, lane, T, H, c0, dh, nb0, nb1);
}

// ============================================================================================= C ABI
static const size_t kFwdLds = (size_t)2 * TR * LDR * sizeof(bf16) + TR * sizeof(float);
static const size_t kBwdLds = (size_t)(2 * TR * LDR + 4 * TR * LDP) * sizeof(bf16) + TR * sizeof(float);

static int x3_args(AttnX3Args& a, int B, int T, int H, int heads, const AderDrop* drop) {
    if (T > TR || heads < 1 || H % heads != 0) return -2;
    const int dh = H / heads;
    if (dh > 160 || (dh & 1)) return -2;
    a.B = B; a.T = T; a.H = H; a.heads = heads; a.sqrt_dh = sqrtf((float)dh);
    a.drop = drop_from(drop);
    return 0;
}

extern "C" {

// PT: [B,heads,T,T] scratch, stored transposed ([key][query]) -- only ader_attn_x3_bwd reads it
int ader_attn_x3_fwd(const float* Q, const float* K, const float* V, const float* q_in, const float* kmask, const float* qmask,
                     float* out, float* PT, int B, int T, int H, int heads, const AderDrop* drop, void* stream) {
    if (B <= 0) return 0;
    AttnX3Args a;
    int rc = x3_args(a, B, T, H, heads, drop);
    if (rc) return rc;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_x3_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    a.Q = Q; a.K = K; a.V = V; a.res = q_in; a.kmask = kmask; a.qmask = qmask; a.out = out; a.PT = PT;
    a.dQ = a.dK = a.dV = nullptr;
    hipLaunchKernelGGL(k_attn_x3_fwd, dim3(B * heads), dim3(128), kFwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_attn_x3_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* PT, const float* kmask,
                     const float* qmask, float* dQ, float* dK, float* dV, int B, int T, int H, int heads, const AderDrop* drop, void* stream) {
    if (B <= 0) return 0;
    AttnX3Args a;
    int rc = x3_args(a, B, T, H, heads, drop);
    if (rc) return rc;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_x3_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    a.Q = Q; a.K = K; a.V = V; a.res = dO; a.kmask = kmask; a.qmask = qmask; a.out = nullptr; a.PT = (float*)PT;
    a.dQ = dQ; a.dK = dK; a.dV = dV;
    hipLaunchKernelGGL(k_attn_x3_bwd, dim3(B * heads), dim3(256), kBwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
