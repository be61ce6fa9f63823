// Causal masked self-attention core of the SASRec block, one workgroup per (sequence, head), all tiles
// LDS-resident, products on v_mfma_f32_16x16x4_f32.  Reference: multihead_attention, modules.py:177-223
//   S = Q K^T / sqrt(dh); key mask (sum_c keys == 0) and causal mask -> -2^32+1; softmax over all T keys;
//   * query mask (sum_c queries != 0); dropout on the probabilities; . V; + queries (the LN'd input).
// The Q/K/V projections (modules.py:172-174) are ader_gemm_rows launches.  T <= 64, dh = H/heads <= 160.
#include "common.h"
#include "../../include/ader_hip.h"

#define TR 64           // padded sequence length (4 MFMA row blocks)
#define LDQ 162         // (row, k) operand reads conflict-free (== 2 mod 32)
#define LDV 176         // (k, n) operand reads conflict-free (== 16 mod 32)
#define LDS_ 68         // score tile row stride (16-B aligned rows)

struct AttnArgs {
    const float* Q; const float* K; const float* V;   // [B,T,H]
    const float* res;                                   // fwd: LN'd queries (residual); bwd: dO
    const float* kmask; const float* qmask;            // [B,T] 1/0
    float* out;                                         // fwd: x1 [B,T,H]
    float* P;                                           // [B,heads,T,T] softmax output (before the query mask)
    float* dQ; float* dK; float* dV;                    // bwd outputs [B,T,H]
    int B, T, H, heads;
    float sqrt_dh;
    DropArgs drop;
};

__device__ __forceinline__ void stage_rows(float* dst, int ld, const float* src, int T, int H, int c0, int dh, int tid, int nthreads) {
    // dst[t][c] = src[t][c0 + c] for t < T, c < dh; zero elsewhere (rows up to TR, cols up to ld)
    for (int i = tid; i < TR * ld; i += nthreads) {
        const int t = i / ld, c = i - t * ld;
        dst[i] = (t < T && c < dh) ? src[(size_t)t * H + c0 + c] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_attn_fwd(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Q_l = smem;
    float* K_l = Q_l + TR * LDQ;
    float* V_l = K_l + TR * LDQ;
    float* S_l = V_l + TR * LDV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.heads, head = blockIdx.x % a.heads;
    const int T = a.T, H = a.H, dh = H / a.heads, c0 = head * dh;
    const size_t base = (size_t)b * T * H;
    stage_rows(Q_l, LDQ, a.Q + base, T, H, c0, dh, tid, 256);
    stage_rows(K_l, LDQ, a.K + base, T, H, c0, dh, tid, 256);
    stage_rows(V_l, LDV, a.V + base, T, H, c0, dh, tid, 256);
    __syncthreads();
    const int m0 = wave * 16;
    {   // S = Q K^T  (wave: 16 query rows x 64 keys)
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<4>(Q_l + m0 * LDQ, LDQ, 1, K_l, 1, LDQ, (dh + 3) >> 2, acc, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = j * 16 + (lane & 15);
            const bool kvalid = (col < T) && (a.kmask[(size_t)b * T + col] != 0.0f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (lane >> 4) * 4 + r;
                float s = acc[j][r] / a.sqrt_dh;                       // modules.py:185
                if (!kvalid) s = NEG_PAD;                              // modules.py:188-193
                if (col > row) s = NEG_PAD;                            // modules.py:196-202
                S_l[row * LDS_ + col] = s;
            }
        }
    }
    __syncthreads();
    {   // softmax over the T keys of each row: 4 lanes per row, 16 columns each
        const int row = m0 + (lane >> 2), sub = lane & 3;
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int col = sub * 16 + i;
            v[i] = S_l[row * LDS_ + col];
            if (col < T) mx = fmaxf(mx, v[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int col = sub * 16 + i;
            v[i] = (col < T) ? expf(v[i] - mx) : 0.0f;
            sum += v[i];
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const bool rvalid = row < T;
        const float qm = rvalid ? a.qmask[(size_t)b * T + row] : 0.0f;      // modules.py:208-211
        float* Pg = a.P + (((size_t)b * a.heads + head) * T + (rvalid ? row : 0)) * T;
        const uint32_t didx0 = (uint32_t)(((size_t)b * a.heads + head) * T + row) * (uint32_t)T;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int col = sub * 16 + i;
            float p = 0.0f;
            if (rvalid && col < T) {
                p = v[i] / sum;
                Pg[col] = p;
                p = drop_apply(a.drop, didx0 + (uint32_t)col, p * qm);   // modules.py:214
            }
            S_l[row * LDS_ + col] = p;
        }
    }
    __syncthreads();
    {   // O = P V ; out = O + queries
        f32x4 acc[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<10>(S_l + m0 * LDS_, LDS_, 1, V_l, LDV, 1, (T + 3) >> 2, acc, lane);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int n = j * 16 + (lane & 15);
            if (n >= dh) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (lane >> 4) * 4 + r;
                if (row >= T) continue;
                const size_t idx = base + (size_t)row * H + c0 + n;
                a.out[idx] = acc[j][r] + a.res[idx];                    // modules.py:223
            }
        }
    }
}

// Backward of the attention core.  Given dO (gradient of the product P_drop.V; the residual path is
// handled by the caller), recompute P_drop from the saved softmax output and the dropout counter, and
// produce dQ, dK, dV.  LDS phases: {dO,V} -> dP ; softmax backward -> dS, P_drop ; dV ; {K} -> dQ ; {Q} -> dK.
__global__ __launch_bounds__(256) void k_attn_bwd(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X1 = smem;                   // dO
    float* X2 = X1 + TR * LDQ;          // V, then K, then Q
    float* T1 = X2 + TR * LDQ;          // dP -> dS
    float* T2 = T1 + TR * LDS_;         // P_drop
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.heads, head = blockIdx.x % a.heads;
    const int T = a.T, H = a.H, dh = H / a.heads, c0 = head * dh;
    const size_t base = (size_t)b * T * H;
    const int m0 = wave * 16;
    const int kd = (dh + 3) >> 2, kt = (T + 3) >> 2;
    stage_rows(X1, LDQ, a.res + base, T, H, c0, dh, tid, 256);
    stage_rows(X2, LDQ, a.V + base, T, H, c0, dh, tid, 256);
    __syncthreads();
    {   // dP_drop[q][key] = sum_c dO[q][c] V[key][c]
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<4>(X1 + m0 * LDQ, LDQ, 1, X2, 1, LDQ, kd, acc, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) T1[(m0 + (lane >> 4) * 4 + r) * LDS_ + j * 16 + (lane & 15)] = acc[j][r];
    }
    __syncthreads();
    {   // softmax backward per row (4 lanes per row)
        const int row = m0 + (lane >> 2), sub = lane & 3;
        const bool rvalid = row < T;
        const float qm = rvalid ? a.qmask[(size_t)b * T + row] : 0.0f;
        const float* Pg = a.P + (((size_t)b * a.heads + head) * T + (rvalid ? row : 0)) * T;
        const uint32_t didx0 = (uint32_t)(((size_t)b * a.heads + head) * T + row) * (uint32_t)T;
        float p[16], dp[16];
        float dot = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int col = sub * 16 + i;
            p[i] = 0.0f; dp[i] = 0.0f;
            float pd = 0.0f;
            if (rvalid && col < T) {
                p[i] = Pg[col];
                float f = qm;                                            // d(p*qm*keep*scale)/dp
                if (a.drop.thr != 0) f = drop_keep(a.drop, didx0 + (uint32_t)col) ? f * a.drop.scale : 0.0f;
                dp[i] = T1[row * LDS_ + col] * f;
                pd = p[i] * f;
                dot += dp[i] * p[i];
            }
            T2[row * LDS_ + col] = pd;
        }
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int col = sub * 16 + i;
            float ds = 0.0f;
            if (rvalid && col < T && col <= row && a.kmask[(size_t)b * T + col] != 0.0f)
                ds = (p[i] * (dp[i] - dot)) / a.sqrt_dh;
            T1[row * LDS_ + col] = ds;
        }
    }
    __syncthreads();
    {   // dV[key][c] = sum_q P_drop[q][key] dO[q][c]
        f32x4 acc[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<10>(T2 + m0, 1, LDS_, X1, LDQ, 1, kt, acc, lane);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int n = j * 16 + (lane & 15);
            if (n >= dh) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (lane >> 4) * 4 + r;
                if (row < T) a.dV[base + (size_t)row * H + c0 + n] = acc[j][r];
            }
        }
    }
    __syncthreads();
    stage_rows(X2, LDQ, a.K + base, T, H, c0, dh, tid, 256);
    __syncthreads();
    {   // dQ[q][c] = sum_key dS[q][key] K[key][c]
        f32x4 acc[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<10>(T1 + m0 * LDS_, LDS_, 1, X2, LDQ, 1, kt, acc, lane);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int n = j * 16 + (lane & 15);
            if (n >= dh) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (lane >> 4) * 4 + r;
                if (row < T) a.dQ[base + (size_t)row * H + c0 + n] = acc[j][r];
            }
        }
    }
    __syncthreads();
    stage_rows(X2, LDQ, a.Q + base, T, H, c0, dh, tid, 256);
    __syncthreads();
    {   // dK[key][c] = sum_q dS[q][key] Q[q][c]
        f32x4 acc[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<10>(T1 + m0, 1, LDS_, X2, LDQ, 1, kt, acc, lane);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int n = j * 16 + (lane & 15);
            if (n >= dh) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (lane >> 4) * 4 + r;
                if (row < T) a.dK[base + (size_t)row * H + c0 + n] = acc[j][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Last-query attention.  Only position T-1 of the final block feeds the representation (ADER.py:85), and rows of a
// block are independent except through K/V, so the final block needs ONE query row per sequence (exact, not an
// approximation; the reference computes all T rows and discards T-1 of them).  Tiny: no MFMA.
// One workgroup per (sequence, head); K and V tiles staged in LDS.
struct AttnLastArgs {
    const float* Ql;            // [B,H]  query projection of row T-1
    const float* K; const float* V;       // [B,T,H]
    const float* res;           // fwd: LN'd query row [B,H];  bwd: dO [B,H]
    const float* kmask;         // [B,T]
    const float* qmask;         // [B]
    float* out;                 // fwd: x1 row T-1 [B,H]
    float* P;                   // [B,heads,T] softmax output of the last query row
    float* dQl; float* dK; float* dV;     // bwd: [B,H], [B,T,H], [B,T,H]
    int B, T, H, heads;
    float sqrt_dh;
    DropArgs drop;
};

#define LAST_LD 161             // odd row stride: per-row dot products read LDS conflict-free

__global__ __launch_bounds__(256) void k_attn_last_fwd(AttnLastArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* K_l = smem;                       // [TR][LAST_LD]
    float* V_l = K_l + TR * LAST_LD;
    float* q_l = V_l + TR * LAST_LD;         // [160]
    float* p_l = q_l + 160;                  // [TR]
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.x / a.heads, head = blockIdx.x % a.heads;
    const int T = a.T, H = a.H, dh = H / a.heads, c0 = head * dh;
    const size_t base = (size_t)b * T * H;
    for (int i0 = 0; i0 < T * dh; i0 += 256 * 8) {          // 16 independent loads in flight per thread (latency-bound staging)
        float kv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + tid + 256 * u;
            const int t = i / dh, c = i - t * dh;
            const bool ok = i < T * dh;
            kv[u] = ok ? a.K[base + (size_t)t * H + c0 + c] : 0.0f;
            vv[u] = ok ? a.V[base + (size_t)t * H + c0 + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + tid + 256 * u;
            const int t = i / dh, c = i - t * dh;
            if (i < T * dh) { K_l[t * LAST_LD + c] = kv[u]; V_l[t * LAST_LD + c] = vv[u]; }
        }
    }
    for (int c = tid; c < dh; c += 256) q_l[c] = a.Ql[(size_t)b * H + c0 + c];
    __syncthreads();
    if (tid < 64) {                           // wave 0: scores + softmax of the single query row
        float s = -INFINITY;
        if (lane < T) {
            float acc = 0.0f;
            for (int c = 0; c < dh; ++c) acc = fmaf(q_l[c], K_l[lane * LAST_LD + c], acc);
            s = acc / a.sqrt_dh;
            if (a.kmask[(size_t)b * T + lane] == 0.0f) s = NEG_PAD;      // causal mask admits every key for row T-1
        }
        const float mx = wave_max(s);
        const float e = (lane < T) ? expf(s - mx) : 0.0f;
        const float sum = wave_sum(e);
        if (lane < T) {
            const float p = e / sum;
            a.P[((size_t)b * a.heads + head) * T + lane] = p;
            const uint32_t didx = (uint32_t)(((size_t)b * a.heads + head) * T + (T - 1)) * (uint32_t)T + (uint32_t)lane;
            p_l[lane] = drop_apply(a.drop, didx, p * a.qmask[b]);
        }
    }
    __syncthreads();
    for (int c = tid; c < dh; c += 256) {
        float acc = 0.0f;
        for (int t = 0; t < T; ++t) acc = fmaf(p_l[t], V_l[t * LAST_LD + c], acc);
        const size_t idx = (size_t)b * H + c0 + c;
        a.out[idx] = acc + a.res[idx];
    }
}

// Backward of the single query row T-1 (pruned last block).  Row layout: 16 lanes per key row (lane sub = column sub + 16 i), four
// rows per wave, four passes: every V and K element of the session is requested up front (80 independent loads per lane, one
// memory round trip), the per-row dot products are 16-lane DPP sums, and the only LDS traffic is the [T] softmax vectors and the
// 16 row-group partials of dQ.  (The first version staged K and V through LDS with one wave doing the dot products serially:
// 36 us per launch at B = 512 against ~12 us of HBM time.)
#define DPPF(v_, ctrl_) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v_)), (ctrl_), 0xf, 0xf, false))
__device__ __forceinline__ float sum16(float v) {
    v += DPPF(v, 0xB1); v += DPPF(v, 0x4E); v += DPPF(v, 0x141); v += DPPF(v, 0x140);
    return v;
}
__global__ __launch_bounds__(256) void k_attn_last_bwd(AttnLastArgs a) {
    __shared__ float acc_l[TR], pd_l[TR], ds_l[TR];
    __shared__ float part_l[16][160];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, rsub = lane >> 4, grp = 4 * wave + rsub;        // row group 0..15: rows grp, grp + 16, ...
    const int b = blockIdx.x / a.heads, head = blockIdx.x % a.heads;
    const int T = a.T, H = a.H, dh = H / a.heads, c0 = head * dh;
    const size_t base = (size_t)b * T * H + c0;
    float g[10], q[10], v[4][10], kk[4][10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int c = sub + 16 * i;
        g[i] = (c < dh) ? a.res[(size_t)b * H + c0 + c] : 0.0f;
        q[i] = (c < dh) ? a.Ql[(size_t)b * H + c0 + c] : 0.0f;
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int t = grp + 16 * ps;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = sub + 16 * i;
            const bool ok = t < T && c < dh;
            v[ps][i] = ok ? a.V[base + (size_t)t * H + c] : 0.0f;
            kk[ps][i] = ok ? a.K[base + (size_t)t * H + c] : 0.0f;
        }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 10; ++i) s = fmaf(g[i], v[ps][i], s);
        s = sum16(s);
        const int t = grp + 16 * ps;
        if (sub == 0 && t < TR) acc_l[t] = s;
    }
    __syncthreads();
    if (tid < 64) {
        float p = 0.0f, dp = 0.0f, pd = 0.0f;
        if (lane < T) {
            p = a.P[((size_t)b * a.heads + head) * T + lane];
            float f = a.qmask[b];
            const uint32_t didx = (uint32_t)(((size_t)b * a.heads + head) * T + (T - 1)) * (uint32_t)T + (uint32_t)lane;
            if (a.drop.thr != 0) f = drop_keep(a.drop, didx) ? f * a.drop.scale : 0.0f;
            dp = acc_l[lane] * f;
            pd = p * f;
        }
        const float dot = wave_sum(dp * p);
        pd_l[lane] = pd;
        ds_l[lane] = (lane < T && a.kmask[(size_t)b * T + lane] != 0.0f) ? (p * (dp - dot)) / a.sqrt_dh : 0.0f;
    }
    __syncthreads();
    float dq[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) dq[i] = 0.0f;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int t = grp + 16 * ps;
        if (t < T) {
            const float pd = pd_l[t], ds = ds_l[t];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int c = sub + 16 * i;
                if (c < dh) {
                    a.dV[base + (size_t)t * H + c] = pd * g[i];
                    a.dK[base + (size_t)t * H + c] = ds * q[i];
                }
                dq[i] = fmaf(ds, kk[ps][i], dq[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) part_l[grp][sub + 16 * i] = dq[i];
    __syncthreads();
    if (tid < dh) {
        float s = 0.0f;
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) s += part_l[gi][tid];       // fixed order: bit-reproducible
        a.dQl[(size_t)b * H + c0 + tid] = s;
    }
}

// ============================================================================================= C ABI
static const size_t kAttnFwdLds = (size_t)(2 * TR * LDQ + TR * LDV + TR * LDS_) * sizeof(float);
static const size_t kAttnBwdLds = (size_t)(2 * TR * LDQ + 2 * TR * LDS_) * sizeof(float);

extern "C" {

int ader_attn_fwd(const float* Q, const float* K, const float* V, const float* q_in, const float* kmask, const float* qmask,
                  float* out, float* P, int B, int T, int H, int heads, const AderDrop* drop, void* stream) {
    if (B <= 0) return 0;
    if (T > TR || heads < 1 || H % heads != 0 || H / heads > 160) return -2;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttnFwdLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    AttnArgs a;
    a.Q = Q; a.K = K; a.V = V; a.res = q_in; a.kmask = kmask; a.qmask = qmask; a.out = out; a.P = P;
    a.dQ = a.dK = a.dV = nullptr;
    a.B = B; a.T = T; a.H = H; a.heads = heads; a.sqrt_dh = sqrtf((float)(H / heads));
    a.drop = drop_from(drop);
    hipLaunchKernelGGL(k_attn_fwd, dim3(B * heads), dim3(256), kAttnFwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_attn_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* P, const float* kmask,
                  const float* qmask, float* dQ, float* dK, float* dV, int B, int T, int H, int heads, const AderDrop* drop, void* stream) {
    if (B <= 0) return 0;
    if (T > TR || heads < 1 || H % heads != 0 || H / heads > 160) return -2;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttnBwdLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    AttnArgs a;
    a.Q = Q; a.K = K; a.V = V; a.res = dO; a.kmask = kmask; a.qmask = qmask; a.out = nullptr; a.P = (float*)P;
    a.dQ = dQ; a.dK = dK; a.dV = dV;
    a.B = B; a.T = T; a.H = H; a.heads = heads; a.sqrt_dh = sqrtf((float)(H / heads));
    a.drop = drop_from(drop);
    hipLaunchKernelGGL(k_attn_bwd, dim3(B * heads), dim3(256), kAttnBwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

static const size_t kAttnLastLds = (size_t)(2 * TR * LAST_LD + 2 * 160 + 2 * TR) * sizeof(float);

static int attn_last_args(AttnLastArgs& a, int B, int T, int H, int heads, const AderDrop* drop) {
    if (T > TR || heads < 1 || H % heads != 0 || H / heads > 160) return -2;
    a.B = B; a.T = T; a.H = H; a.heads = heads; a.sqrt_dh = sqrtf((float)(H / heads));
    a.drop = drop_from(drop);
    return 0;
}

int ader_attn_last_fwd(const float* Q_last, const float* K, const float* V, const float* q_in_last, const float* kmask,
                       const float* qmask_last, float* out_last, float* P_last, int B, int T, int H, int heads, const AderDrop* drop, void* stream) {
    if (B <= 0) return 0;
    AttnLastArgs a;
    int rc = attn_last_args(a, B, T, H, heads, drop);
    if (rc) return rc;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_last_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAttnLastLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    a.Ql = Q_last; a.K = K; a.V = V; a.res = q_in_last; a.kmask = kmask; a.qmask = qmask_last; a.out = out_last; a.P = P_last;
    a.dQl = a.dK = a.dV = nullptr;
    hipLaunchKernelGGL(k_attn_last_fwd, dim3(B * heads), dim3(256), kAttnLastLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_attn_last_bwd(const float* dO_last, const float* Q_last, const float* K, const float* V, const float* P_last,
                       const float* kmask, const float* qmask_last, float* dQ_last, float* dK, float* dV, int B, int T, int H,
                       int heads, const AderDrop* drop, void* stream) {
    if (B <= 0) return 0;
    AttnLastArgs a;
    int rc = attn_last_args(a, B, T, H, heads, drop);
    if (rc) return rc;
    a.Ql = Q_last; a.K = K; a.V = V; a.res = dO_last; a.kmask = kmask; a.qmask = qmask_last; a.out = nullptr; a.P = (float*)P_last;
    a.dQl = dQ_last; a.dK = dK; a.dV = dV;
    hipLaunchKernelGGL(k_attn_last_bwd, dim3(B * heads), dim3(256), 0, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
