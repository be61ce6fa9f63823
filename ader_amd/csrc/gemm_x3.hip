// Block GEMMs on the bf16 matrix cores at float32-grade accuracy ("bf16x3"): every fp32 operand x is split into
// hi = bf16(x), lo = bf16(x - hi) and a product is evaluated as hi.hi + lo.hi + hi.lo with fp32 accumulation
// (error ~2^-16 relative per product; the dropped lo.lo term is ~2^-18).  v_mfma_f32_32x32x16_bf16 runs 16x the rate
// of the f32 MFMA, so three of them are still > 5x faster and the kernels become bound by moving the activations
// through the Infinity Cache rather than by the matrix pipe.
//
// Same operations / epilogues as gemm.hip (reference modules.py:172-174 dense Q/K/V, modules.py:254-261 conv1d(k=1)
// FFN, and their backward products):
//   ader_wprep       : W [H,H] fp32 -> fragment-ready bf16 (hi, lo) copies of W^T and W, rows of 168 elements
//   ader_gemm_x3     : C[M,H] = epilogue(A[M,H] . W (+bias))   (trans_b: A . W^T)
//   ader_gemm_atb_x3 : dW = A^T . G, db = colsum(G)   (operands read k-major from row-major LDS tiles with
//                      ds_read_b64_tr_b16; per-workgroup slabs, deterministic reduce)
// H <= 159 (ones-column bias trick), H even.  gfx950 only.
#include "lbf_common.h"
#include "x3_image.h"
#include <string.h>
#include <stdlib.h>
#include "../../include/ader_hip.h"

#define TM 64
#define WSZ (HP * LDR)            // elements of one prepared weight plane

enum { EPI_BIAS = 0, EPI_BIAS_RELU_DROP = 1, EPI_BIAS_DROP_RES_MASK = 2, EPI_RELUDROPGRAD = 3, EPI_ADD = 4 };

__device__ __forceinline__ void split2(float x, float y, bf16x2& hi, bf16x2& lo) {
    hi[0] = (bf16)x; hi[1] = (bf16)y;
    lo[0] = (bf16)(x - (float)hi[0]); lo[1] = (bf16)(y - (float)hi[1]);
}

// planes per weight: [0] W^T hi, [1] W^T lo, [2] W hi, [3] W lo; a plane holds B[n][k] (zero padded to 160 x 160) in FRAGMENT order:
// element (n = 32 nb + r, k = 16 ks + 8 hh + j) at ((nb * 10 + ks) * 64 + 32 hh + r) * 8 + j, i.e. the 64 lanes' 16-byte operands of one
// v_mfma_f32_32x32x16_bf16 B fragment are 1 KiB contiguous: a wave's fragment load touches 8 fully used cache lines (the row-major
// [n][168] planes of rounds 1-2 made every such load touch 32 lines at 25 % use; the ten waves' 20 KB streams overflowed the 32 KB
// L1 and every line was fetched from L2 several times: 1 GB per k_seq_fwd launch).
__device__ __host__ __forceinline__ constexpr int wfrag_off(int nb, int ks, int lane) { return ((nb * 10 + ks) * 64 + lane) * 8; }
__global__ __launch_bounds__(256) void k_wprep(const float* __restrict__ theta, const long* __restrict__ offs, int nw, int H,
                                               bf16* __restrict__ out) {
    const int w = blockIdx.y;
    if (w >= nw) return;
    const float* W = theta + offs[w];
    bf16* o = out + (size_t)w * 4 * WSZ;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < WSZ; i += gridDim.x * blockDim.x) {
        float vt = 0.0f, vn = 0.0f;
        if (i < 5 * 10 * 64 * 8) {
            const int j = i & 7, lane = (i >> 3) & 63, f = i >> 9;
            const int ks = f % 10, nb = f / 10;
            const int n = 32 * nb + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
            if (n < H && k < H) { vt = W[(size_t)k * H + n]; vn = W[(size_t)n * H + k]; }     // W^T[n][k], W[n][k]
        }
        const bf16 th = (bf16)vt, nh = (bf16)vn;
        o[i] = th; o[WSZ + i] = (bf16)(vt - (float)th);
        o[2 * WSZ + i] = nh; o[3 * WSZ + i] = (bf16)(vn - (float)nh);
    }
}

struct GemmX3Args {
    const float* A; const bf16* Bhi; const bf16* Blo; const float* bias; float* C;
    const float* aux; const int* seq;
    int M, H, row_mul, row_add;
    DropArgs drop;
};

#define PFG 8      // float2 prefetch registers per thread (64*80/640 = 8)

// 10 waves: wave w -> output columns 32*(w%5).., rows 32*(w/5).. of the 64-row tile.  B fragments live in registers.
template <int EPI>
__global__ __launch_bounds__(640) void k_gemm_x3(GemmX3Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* A_l = (bf16*)smem_raw;                 // [2 buffers][2 (hi,lo)][TM][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nb = wave % 5, mh = wave / 5;
    const int H = g.H, HH = H >> 1, M = g.M;
    bf16x8 bh[10], bl[10];
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        bh[ks] = *(const bf16x8*)(g.Bhi + wfrag_off(nb, ks, lane));
        bl[ks] = *(const bf16x8*)(g.Blo + wfrag_off(nb, ks, lane));
    }
    for (int i = tid; i < 2 * 2 * TM * LDR / 2; i += 640) ((uint32_t*)A_l)[i] = 0u;     // K padding columns stay zero
    const int it_first = tid / HH, c2_first = tid - it_first * HH;
    const int it_step = 640 / HH, c2_step = 640 - it_step * HH;
    float2 pf[PFG];
#define GX_PREFETCH(m0_)                                                                                 \
    {                                                                                                    \
        int it_ = it_first, c2_ = c2_first;                                                              \
        _Pragma("unroll") for (int j = 0; j < PFG; ++j) {                                                \
            float2 v_ = make_float2(0.f, 0.f);                                                           \
            if (it_ < TM && (m0_) + it_ < M) v_ = *(const float2*)(g.A + (size_t)((m0_) + it_) * H + 2 * c2_); \
            pf[j] = v_;                                                                                  \
            it_ += it_step; c2_ += c2_step;                                                              \
            if (c2_ >= HH) { c2_ -= HH; ++it_; }                                                         \
        }                                                                                                \
    }
#define GX_STAGE(buf_)                                                                                   \
    {                                                                                                    \
        bf16* hi_ = A_l + (buf_) * 2 * TM * LDR;                                                         \
        bf16* lo_ = hi_ + TM * LDR;                                                                      \
        int it_ = it_first, c2_ = c2_first;                                                              \
        _Pragma("unroll") for (int j = 0; j < PFG; ++j) {                                                \
            if (it_ < TM) {                                                                              \
                bf16x2 h_, l_;                                                                           \
                split2(pf[j].x, pf[j].y, h_, l_);                                                        \
                *(bf16x2*)(hi_ + it_ * LDR + 2 * c2_) = h_;                                              \
                *(bf16x2*)(lo_ + it_ * LDR + 2 * c2_) = l_;                                              \
            }                                                                                            \
            it_ += it_step; c2_ += c2_step;                                                              \
            if (c2_ >= HH) { c2_ -= HH; ++it_; }                                                         \
        }                                                                                                \
    }
    const int n_tiles = (M + TM - 1) / TM;
    __syncthreads();
    int tile = blockIdx.x;
    if (tile < n_tiles) { GX_PREFETCH(tile * TM); GX_STAGE(0); }
    __syncthreads();
    int cur = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        const bool more = tile + gridDim.x < n_tiles;
        if (more) GX_PREFETCH((tile + gridDim.x) * TM);
        const bf16* Ah = A_l + cur * 2 * TM * LDR + (32 * mh + r) * LDR + 8 * hh;
        const bf16* Al = Ah + TM * LDR;
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            const bf16x8 ah = *(const bf16x8*)(Ah + 16 * ks);
            const bf16x8 al = *(const bf16x8*)(Al + 16 * ks);
            acc = mfma_bf16(al, bh[ks], acc);
            acc = mfma_bf16(ah, bl[ks], acc);
            acc = mfma_bf16(ah, bh[ks], acc);
        }
        const int n = 32 * nb + r;
        if (n < H) {
            const float bv = (EPI <= EPI_BIAS_DROP_RES_MASK && g.bias) ? g.bias[n] : 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int m = m0 + 32 * mh + acc_row(j, hh);
                if (m >= M) continue;
                const size_t idx = (size_t)m * H + n;
                const int mf = m * g.row_mul + g.row_add;
                const uint32_t didx = (uint32_t)mf * (uint32_t)H + (uint32_t)n;
                float v = acc[j] + bv;
                if (EPI == EPI_BIAS_RELU_DROP) {
                    v = fmaxf(v, 0.0f);
                    v = drop_apply(g.drop, didx, v);
                } else if (EPI == EPI_BIAS_DROP_RES_MASK) {
                    v = drop_apply(g.drop, didx, v);
                    v = (g.seq[mf] != 0) ? (v + g.aux[idx]) : 0.0f;
                } else if (EPI == EPI_RELUDROPGRAD) {
                    v = (g.aux[idx] != 0.0f) ? v * g.drop.scale : 0.0f;
                } else if (EPI == EPI_ADD) {
                    v = v + g.aux[idx];
                }
                g.C[idx] = v;
            }
        }
        if (more) GX_STAGE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
}

#define PFA 15     // float2 prefetch registers per operand per thread (64*75/320 = 15) ... H <= 150 here

// dW_aug slab [HP][HP] per workgroup: rows = input channel (row H = ones column -> bias gradient), cols = output channel.
// 5 waves: wave w owns input channels 32w..32w+31 and all 5 output-channel blocks (80 accumulator registers).
__device__ __forceinline__ void atb_x3_body(const float* __restrict__ A, const float* __restrict__ G, float* __restrict__ out,
                                            int M, int H, int wg, int nwg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* Ah = (bf16*)smem_raw;                  // [TM][LDR] each
    bf16* Al = Ah + TM * LDR;
    bf16* Gh = Al + TM * LDR;
    bf16* Gl = Gh + TM * LDR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int HH = H >> 1;
    f32x16 acc[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[nb][j] = 0.0f;
    for (int i = tid; i < 4 * TM * LDR / 2; i += 320) ((uint32_t*)Ah)[i] = 0u;
    const int it_first = tid / HH, c2_first = tid - it_first * HH;
    const int it_step = 320 / HH, c2_step = 320 - it_step * HH;
    float2 pa[PFA], pg[PFA];
    const int n_tiles = (M + TM - 1) / TM;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    __syncthreads();
    int tile = wg;
#define AT_PREFETCH(m0_)                                                                                 \
    {                                                                                                    \
        int it_ = it_first, c2_ = c2_first;                                                              \
        _Pragma("unroll") for (int j = 0; j < PFA; ++j) {                                                \
            float2 va_ = make_float2(0.f, 0.f), vg_ = make_float2(0.f, 0.f);                             \
            if (it_ < TM && (m0_) + it_ < M) {                                                           \
                va_ = *(const float2*)(A + (size_t)((m0_) + it_) * H + 2 * c2_);                         \
                vg_ = *(const float2*)(G + (size_t)((m0_) + it_) * H + 2 * c2_);                         \
            }                                                                                            \
            pa[j] = va_; pg[j] = vg_;                                                                    \
            it_ += it_step; c2_ += c2_step;                                                              \
            if (c2_ >= HH) { c2_ -= HH; ++it_; }                                                         \
        }                                                                                                \
    }
#define AT_STAGE(m0_)                                                                                    \
    {                                                                                                    \
        int it_ = it_first, c2_ = c2_first;                                                              \
        _Pragma("unroll") for (int j = 0; j < PFA; ++j) {                                                \
            if (it_ < TM) {                                                                              \
                bf16x2 h_, l_;                                                                           \
                split2(pa[j].x, pa[j].y, h_, l_);                                                        \
                *(bf16x2*)(Ah + it_ * LDR + 2 * c2_) = h_; *(bf16x2*)(Al + it_ * LDR + 2 * c2_) = l_;    \
                split2(pg[j].x, pg[j].y, h_, l_);                                                        \
                *(bf16x2*)(Gh + it_ * LDR + 2 * c2_) = h_; *(bf16x2*)(Gl + it_ * LDR + 2 * c2_) = l_;    \
            }                                                                                            \
            it_ += it_step; c2_ += c2_step;                                                              \
            if (c2_ >= HH) { c2_ -= HH; ++it_; }                                                         \
        }                                                                                                \
        if (tid < TM) Ah[tid * LDR + H] = (bf16)(((m0_) + tid < M) ? 1.0f : 0.0f);   /* ones column -> db */ \
    }
    if (tile < n_tiles) AT_PREFETCH(tile * TM);
    for (; tile < n_tiles; tile += nwg) {
        __syncthreads();                                    // previous tile's reads are done
        AT_STAGE(tile * TM);
        __syncthreads();
        if (tile + nwg < n_tiles) AT_PREFETCH((tile + nwg) * TM);
#pragma unroll
        for (int ks = 0; ks < TM / 16; ++ks) {
            const int ro = (16 * ks + 4 * hh + q4) * LDR + 16 * g1 + 4 * p4;
            bf16x8 ah, al;
            {
                const bf16x4 x0 = tr_read(Ah + ro + 32 * wave), x1 = tr_read(Ah + ro + 32 * wave + 8 * LDR);
                const bf16x4 y0 = tr_read(Al + ro + 32 * wave), y1 = tr_read(Al + ro + 32 * wave + 8 * LDR);
#pragma unroll
                for (int j = 0; j < 4; ++j) { ah[j] = x0[j]; ah[4 + j] = x1[j]; al[j] = y0[j]; al[4 + j] = y1[j]; }
            }
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const bf16x4 x0 = tr_read(Gh + ro + 32 * nb), x1 = tr_read(Gh + ro + 32 * nb + 8 * LDR);
                const bf16x4 y0 = tr_read(Gl + ro + 32 * nb), y1 = tr_read(Gl + ro + 32 * nb + 8 * LDR);
                bf16x8 gh, gl;
#pragma unroll
                for (int j = 0; j < 4; ++j) { gh[j] = x0[j]; gh[4 + j] = x1[j]; gl[j] = y0[j]; gl[4 + j] = y1[j]; }
                acc[nb] = mfma_bf16(al, gh, acc[nb]);
                acc[nb] = mfma_bf16(ah, gl, acc[nb]);
                acc[nb] = mfma_bf16(ah, gh, acc[nb]);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) out[(size_t)(32 * wave + acc_row(j, hh)) * HP + 32 * nb + r] = acc[nb][j];
}

__global__ __launch_bounds__(320) void k_gemm_atb_x3(const float* __restrict__ A, const float* __restrict__ G, float* __restrict__ slab,
                                                      int M, int H) {
    atb_x3_body(A, G, slab + (size_t)blockIdx.x * HP * HP, M, H, blockIdx.x, gridDim.x);
}

// All weight-gradient products of a backward pass in ONE launch (they only depend on saved activations and on the
// gradient rows the backward chain has already written): workgroup w serves product y with wg0[y] <= w < wg0[y+1].
#define ATB_MAX 16
struct AtbBatch {
    const float* A[ATB_MAX]; const float* G[ATB_MAX]; float* dW[ATB_MAX]; float* db[ATB_MAX];
    int M[ATB_MAX]; int wg0[ATB_MAX + 1]; int n;
    // packed session tiles (seqp_*.hip): the operands of product y are in tile order -- 64-row tiles of which tile u holds trows[y][u]
    // rows (the rest was never written) -- and *Mdev[y] = 64 x the number of tiles bounds M[y], the host's upper bound.  NULL: plain rows
    const int* Mdev[ATB_MAX]; const int* trows[ATB_MAX];
};
// dW[y] = sum of product y's slabs (fixed order), row H of the augmented slab -> db[y].  grid (blocks of 64 outputs, n).
__global__ __launch_bounds__(256) void k_atb_reduce_batch(AtbBatch b, const float* __restrict__ slab, int H) {
    __shared__ float red[16][16];     // 16 slab groups x 16 outputs: small workgroups that fit next to the table update
    const int y = blockIdx.y;
    const int S = b.wg0[y + 1] - b.wg0[y];
    const float* src = slab + (size_t)b.wg0[y] * HP * HP;
    const int total = (H + 1) * H;
    const int o = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    float acc = 0.0f;
    int r = 0, c = 0;
    if (i < total) {
        r = i / H; c = i - r * H;
        const float* p = src + (size_t)r * HP + c;
#pragma unroll 4
        for (int s = sg; s < S; s += 16) acc += p[(size_t)s * HP * HP];
    }
    red[sg][o] = acc;
    __syncthreads();
    if (sg == 0 && i < total) {
        float a = red[0][o];
#pragma unroll
        for (int k = 1; k < 16; ++k) a += red[k][o];
        if (r < H) b.dW[y][(size_t)r * H + c] = a;
        else if (b.db[y]) b.db[y][c] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Small-footprint form of the batched weight-gradient products: the one that runs INSIDE the fused table update.
// The batched launch of atb_x3_body (86 KB of LDS, 256 registers x 5 waves; removed in round 4) could not share a CU with update
// workgroups (three per CU, 168 registers, ~47 KB each), so it started when the update drained: a ~0.1 ms tail on every step.  This form is cut to the
// hole ONE retiring update workgroup leaves: 256 threads, <= 168 registers, 43.5 KB of LDS, so the high-priority side stream gets
// its workgroups placed as update workgroups retire.  32-row tiles (the K of one v_mfma_f32_16x16x32_bf16), operands stored as the
// conflict-free hi / lo images of x3_image.h (both operands are read k-major: ds_read_b64_tr_b16), wave w owns output-channel
// row blocks {2w, 2w+1} x 10 column blocks + half of row block 8 + (w >> 1): 25 blocks of 16x16 = 100 accumulator registers.
// No register prefetch across the MFMA phase (the accumulators leave no room): its latency is covered by the update workgroups
// on the same CU -- this is filler work, not a kernel that has the chip to itself.  Slabs and reduce: as the large form.
#define SM_TM 32
struct __attribute__((packed, aligned(8))) AtbVec { f32x4v v; };      // 16-byte vector at an 8-byte aligned address
#define SM_LDS (4 * X3_PLANE_B)
template <int HT>       // HT: hidden size known at compile time (150: the reference default, main.py:104) or 0 = runtime
__global__ __launch_bounds__(256, 3) void k_gemm_atb_x3_sm(AtbBatch b, float* __restrict__ slab, int Hrt) {
    const int H = HT ? HT : Hrt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int y = 0;
#pragma unroll 1
    while (y + 1 < b.n && (int)blockIdx.x >= b.wg0[y + 1]) ++y;
    const float* __restrict__ A = b.A[y];
    const float* __restrict__ G = b.G[y];
    int M = b.M[y];
    const int* __restrict__ trows = b.trows[y];
    if (b.Mdev[y]) M = min(M, *b.Mdev[y]);
    const int wg = blockIdx.x - b.wg0[y], nwg = b.wg0[y + 1] - b.wg0[y];
    float* __restrict__ out = slab + (size_t)blockIdx.x * HP * HP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3;
    unsigned char* Ah = smem_raw;                      // planes: A hi, A lo, G hi, G lo
    for (int i = tid; i < SM_LDS / 16; i += 256) ((uint4*)smem_raw)[i] = make_uint4(0u, 0u, 0u, 0u);
    f32x4v acc[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) acc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const int nkc = (H + 8) >> 3;                      // k-chunks that hold channels 0..H (channel H of A: the ones column -> db)
    const int nslot = SM_TM * nkc;
    const int n_tiles = (M + SM_TM - 1) / SM_TM;
    // transposed read of (rows 4g + q4 and 16 + 4g + q4, 16-channel block cb): see x3_image.h
    const int t_off = 1152 * (p4 >> 1) + 16 * (4 * g + q4) + 8 * (p4 & 1);
    const int rb2 = 8 + (wave >> 1), cb2 = 5 * (wave & 1);
    __syncthreads();
    for (int tile = wg; tile < n_tiles; tile += nwg) {
        const int m0 = tile * SM_TM;
        int mlim = M - m0;                                 // rows of this 32-row tile that exist
        if (trows) {
            mlim = min(mlim, trows[m0 >> 6] - (m0 & 63));
            if (mlim <= 0) continue;                       // (workgroup-uniform: the second half of a session tile with <= 32 rows)
        }
        // ---- stage: slot s = (row m = s / nkc, k-chunk kc = s % nkc): consecutive lanes read consecutive 32-byte pieces of a row
#pragma unroll
        for (int op = 0; op < 2; ++op) {               // one operand at a time: 40 staging registers beside the 100 accumulators
            const float* __restrict__ src = op ? G : A;
            float2 v[5][4];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int s = tid + 256 * r;
                const int m = s / nkc, kc = s - m * nkc;
                const bool rowok = s < nslot && m < mlim;
                const float* ps = src + (size_t)(m0 + m) * H + 8 * kc;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[r][j] = make_float2(0.f, 0.f);
                if (rowok && 8 * kc + 8 <= H) {            // full k-chunk: two 16-byte loads (rows are 8-byte aligned: H even)
                    const AtbVec q0 = *(const AtbVec*)ps, q1 = *(const AtbVec*)(ps + 4);
                    v[r][0] = make_float2(q0.v[0], q0.v[1]); v[r][1] = make_float2(q0.v[2], q0.v[3]);
                    v[r][2] = make_float2(q1.v[0], q1.v[1]); v[r][3] = make_float2(q1.v[2], q1.v[3]);
                } else if (rowok) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (8 * kc + 2 * j < H) v[r][j] = *(const float2*)(ps + 2 * j);
                }
            }
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int s = tid + 256 * r;
                const int m = s / nkc, kc = s - m * nkc;
                if (s < nslot) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { x[2 * j] = v[r][j].x; x[2 * j + 1] = v[r][j].y; }
                    if (op == 0 && kc == (H >> 3) && m < mlim) x[H & 7] = 1.0f;          // ones column
                    bf16x8 h_, l_;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { h_[j] = (bf16)x[j]; l_[j] = (bf16)(x[j] - (float)h_[j]); }
                    const int so = x3_kc_off(kc) + 16 * m + 2 * op * X3_PLANE_B;
                    *(bf16x8*)(Ah + so) = h_;
                    *(bf16x8*)(Ah + X3_PLANE_B + so) = l_;
                }
            }
        }
        __syncthreads();
        // ---- products
#define SM_FRAG(dst_, plane_, blk_)                                                                      \
        { const bf16* tp_ = (const bf16*)(Ah + (plane_) * X3_PLANE_B + t_off + X3_QUAD * ((blk_) >> 1) + 512 * ((blk_) & 1)); \
          const bf16x4 x0_ = tr_read(tp_), x1_ = tr_read(tp_ + 128);                                     \
          _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { dst_[j_] = x0_[j_]; dst_[4 + j_] = x1_[j_]; } }
#pragma unroll
        for (int rbi = 0; rbi < 3; ++rbi) {
            const int rb = rbi < 2 ? 2 * wave + rbi : rb2;
            bf16x8 ah, al;
            SM_FRAG(ah, 0, rb);
            SM_FRAG(al, 1, rb);
#pragma unroll
            for (int cbi = 0; cbi < (rbi < 2 ? 10 : 5); ++cbi) {
                const int cb = rbi < 2 ? cbi : cb2 + cbi;
                bf16x8 gh, gl;
                SM_FRAG(gh, 2, cb);
                SM_FRAG(gl, 3, cb);
                f32x4v& c = acc[rbi < 2 ? 10 * rbi + cbi : 20 + cbi];
                c = mfma16_bf16(al, gh, c);
                c = mfma16_bf16(ah, gl, c);
                c = mfma16_bf16(ah, gh, c);
                __builtin_amdgcn_sched_barrier(0);         // keep hipcc from hoisting all fragment reads of a row block (registers)
            }
        }
        __syncthreads();                                   // the tile's reads are done before the next one is stored
    }
    // ---- slab: out[input channel][output channel]; C layout: column = lane & 15 (G's block), rows 4 (lane >> 4) + i (A's block)
#pragma unroll
    for (int rbi = 0; rbi < 3; ++rbi) {
        const int rb = rbi < 2 ? 2 * wave + rbi : rb2;
#pragma unroll
        for (int cbi = 0; cbi < (rbi < 2 ? 10 : 5); ++cbi) {
            const int cb = rbi < 2 ? cbi : cb2 + cbi;
            const f32x4v c = acc[rbi < 2 ? 10 * rbi + cbi : 20 + cbi];
            const int col = x3_channel(cb, c16);
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(size_t)x3_channel(rb, 4 * g + i) * HP + col] = c[i];
        }
    }
}

// ============================================================================================= C ABI
static const size_t kGemmX3Lds = (size_t)2 * 2 * TM * LDR * sizeof(bf16);
static const size_t kAtbX3Lds = (size_t)4 * TM * LDR * sizeof(bf16);

template <int EPI>
static int launch_x3(const GemmX3Args& g, hipStream_t st) {
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_x3<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmX3Lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int n_tiles = (g.M + TM - 1) / TM;
    const int per = (n_tiles + 255) / 256;
    const int grid = (n_tiles + per - 1) / per;
    hipLaunchKernelGGL((k_gemm_x3<EPI>), dim3(grid), dim3(640), kGemmX3Lds, st, g);
    HIP_LAUNCH_CHECK();
    return 0;
}

extern "C" {

// Elements (bf16) of the prepared-weight buffer for nw weights; planes per weight: W^T hi, W^T lo, W hi, W lo.
size_t ader_wprep_elems(int nw) { return (size_t)nw * 4 * WSZ; }

// offs[nw] (device, int64): offsets of the [H,H] fp32 weights inside theta.
int ader_wprep(const float* theta, const long* offs, int nw, int H, void* out, void* stream) {
    if (nw <= 0) return 0;
    if (H > HP || (H & 1)) return -2;
    hipLaunchKernelGGL(k_wprep, dim3(32, nw), dim3(256), 0, (hipStream_t)stream, theta, offs, nw, H, (bf16*)out);
    HIP_LAUNCH_CHECK();
    return 0;
}

// wplanes: the 4 prepared planes of this weight (from ader_wprep); trans_b selects W (A . W^T) instead of W^T (A . W).
int ader_gemm_x3(const float* A, const void* wplanes, const float* bias, float* C, const float* aux, const int* seq, int M, int H,
                 int epilogue, int trans_b, int row_mul, int row_add, const AderDrop* drop, void* stream) {
    if (M <= 0) return 0;
    if (H > HP || H < 2 || (H & 1)) return -2;
    GemmX3Args g;
    const bf16* wp = (const bf16*)wplanes + (trans_b ? 2 * WSZ : 0);
    g.A = A; g.Bhi = wp; g.Blo = wp + WSZ; g.bias = bias; g.C = C; g.aux = aux; g.seq = seq; g.M = M; g.H = H;
    g.row_mul = row_mul; g.row_add = row_add;
    g.drop = drop_from(drop);
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case EPI_BIAS: return launch_x3<EPI_BIAS>(g, st);
        case EPI_BIAS_RELU_DROP: return launch_x3<EPI_BIAS_RELU_DROP>(g, st);
        case EPI_BIAS_DROP_RES_MASK: return launch_x3<EPI_BIAS_DROP_RES_MASK>(g, st);
        case EPI_RELUDROPGRAD: return launch_x3<EPI_RELUDROPGRAD>(g, st);
        case EPI_ADD: return launch_x3<EPI_ADD>(g, st);
        default: return -3;
    }
}

// slab: ader_gemm_atb_slabs(M)*160*160 floats.  dW [H,H] and db [H] (may be NULL) are overwritten.
int ader_gemm_atb_x3(const float* A, const float* G, float* slab, float* dW, float* db, int M, int H, void* stream) {
    if (M <= 0) return 0;
    if (H >= HP || H < 2 || (H & 1) || H > 2 * PFA * 5) return -2;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_atb_x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAtbX3Lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int S = ader_gemm_atb_slabs(M);
    hipLaunchKernelGGL(k_gemm_atb_x3, dim3(S), dim3(320), kAtbX3Lds, (hipStream_t)stream, A, G, slab, M, H);
    HIP_LAUNCH_CHECK();
    return ader_reduce_slabs(slab, (long)HP * HP, S, HP, H, H, dW, db, stream);
}

// Batched form: n <= 16 products dW[i] = A[i]^T . G[i] (M[i] rows each), db[i] = colsum(G[i]) (db[i] may be NULL), one
// product launch + one reduce launch.  Host arrays of device pointers.  slab: ader_gemm_atb_batch_slabs(M, n)*160*160
// floats.  Workgroups are shared out in proportion to the rows of each product (about one per CU in total).
static void atb_batch_plan(const int* M, int n, int* wg0) {
    const int tm = SM_TM;
    long tiles_total = 0;
    for (int i = 0; i < n; ++i) tiles_total += (M[i] + tm - 1) / tm;
    wg0[0] = 0;
    for (int i = 0; i < n; ++i) {
        const long t = (M[i] + tm - 1) / tm;
        long s = tiles_total > 256 ? (t * 256 + tiles_total / 2) / tiles_total : t;
        if (s < 1) s = 1;
        if (s > t && t > 0) s = t;
        // balanced: every workgroup of the product gets ceil(t/s) or one fewer tiles
        wg0[i + 1] = wg0[i] + (int)s;
    }
}

int ader_gemm_atb_batch_slabs(const int* M, int n) {
    if (n <= 0 || n > ATB_MAX) return 0;
    int wg0[ATB_MAX + 1];
    atb_batch_plan(M, n, wg0);
    return wg0[n];
}

// ... with operands in the tile order of the packed session kernels (include/ader_hip.h): Mplan[i] = the row count the workgroups
// are shared out by (the host's estimate of the rows that exist; M[i] stays the bound of the addresses), Mdev[i] / trows[i] device
// pointers or NULL per product.  Slabs: ader_gemm_atb_batch_slabs(Mplan, n).
int ader_gemm_atb_x3_batch_pk(const float* const* A, const float* const* G, float* const* dW, float* const* db, const int* M,
                              const int* Mplan, const int* const* Mdev, const int* const* trows, int n, float* slab, int H, void* stream) {
    if (n <= 0) return 0;
    if (n > ATB_MAX) return -2;
    if (H >= HP || H < 2 || (H & 1) || H > 2 * PFA * 5) return -2;
    AtbBatch b;
    for (int i = 0; i < n; ++i) {
        if (M[i] <= 0 || (Mplan && (Mplan[i] <= 0 || Mplan[i] > M[i]))) return -2;
        b.A[i] = A[i]; b.G[i] = G[i]; b.dW[i] = dW[i]; b.db[i] = db[i]; b.M[i] = M[i];
        b.Mdev[i] = Mdev ? Mdev[i] : nullptr; b.trows[i] = trows ? trows[i] : nullptr;
    }
    b.n = n;
    atb_batch_plan(Mplan ? Mplan : M, n, b.wg0);
    if (H == 150) hipLaunchKernelGGL(k_gemm_atb_x3_sm<150>, dim3(b.wg0[n]), dim3(256), SM_LDS, (hipStream_t)stream, b, slab, H);
    else hipLaunchKernelGGL(k_gemm_atb_x3_sm<0>, dim3(b.wg0[n]), dim3(256), SM_LDS, (hipStream_t)stream, b, slab, H);
    HIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_atb_reduce_batch, dim3(((H + 1) * H + 15) / 16, n), dim3(256), 0, (hipStream_t)stream, b, slab, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_gemm_atb_x3_batch(const float* const* A, const float* const* G, float* const* dW, float* const* db, const int* M, int n,
                           float* slab, int H, void* stream) {
    if (n <= 0) return 0;
    if (n > ATB_MAX) return -2;
    if (H >= HP || H < 2 || (H & 1) || H > 2 * PFA * 5) return -2;
    AtbBatch b;
    for (int i = 0; i < n; ++i) {
        if (M[i] <= 0) return -2;
        b.A[i] = A[i]; b.G[i] = G[i]; b.dW[i] = dW[i]; b.db[i] = db[i]; b.M[i] = M[i];
        b.Mdev[i] = nullptr; b.trows[i] = nullptr;
    }
    b.n = n;
    atb_batch_plan(M, n, b.wg0);
    if (H == 150) hipLaunchKernelGGL(k_gemm_atb_x3_sm<150>, dim3(b.wg0[n]), dim3(256), SM_LDS, (hipStream_t)stream, b, slab, H);
    else hipLaunchKernelGGL(k_gemm_atb_x3_sm<0>, dim3(b.wg0[n]), dim3(256), SM_LDS, (hipStream_t)stream, b, slab, H);
    HIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_atb_reduce_batch, dim3(((H + 1) * H + 15) / 16, n), dim3(256), 0, (hipStream_t)stream, b, slab, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
