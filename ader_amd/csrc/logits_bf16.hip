// Full-catalog logits + softmax cross-entropy, bf16-MFMA path (fp32 master table, fp32 accumulation and softmax).
// Reference: ADER.py:88-93 (logits = rep . item_emb^T, one-hot softmax CE) and its gradient.  One-hot rows only
// (vanilla loss / disable_distillation); distilled rows use the float32 path of logits.hip.
//
// Two streaming passes over the table instead of the TF graph's materialised [B,N] logits/softmax/one-hot:
//
//   k_lbf_fwd  "flash" forward: workgroup = (128 batch rows) x (item range).  Per 32-item block
//              S^T = E.rep^T on v_mfma_f32_32x32x16_bf16 (items on the MFMA rows, batch rows on the lanes, so the
//              softmax statistics are lane-local), online max/sum-exp, and the probabilities go straight back into
//              the matrix core as the A operand of  O[b,:] += P^T . E  (no LDS round trip: the accumulator layout
//              is the operand layout; the table block is read k-major with ds_read_b64_tr_b16).
//              O/l is the softmax-weighted mean of item embeddings = d loss / d rep up to the target term, so the
//              backward pass needs no second recompute for dRep.
//   k_lbf_combine  merges the per-range partials: lse, loss, dRep, and the per-row exponent offset for the backward.
//   k_lbf_bwd_de   dE tile = dlogit^T . rep per 128-item workgroup: S = rep.E^T recomputed (batch rows on the MFMA
//              rows), p = exp2(S*log2e + off_b) packed to bf16 in registers and fed back as the A operand of
//              dE[item,:] += P^T . rep with rep read k-major by ds_read_b64_tr_b16.  Each dE row is written once.
//   k_lbf_target_fix  the sparse one-hot term: dE[label_b] -= w_b * rep_b.
//
// LDS tiles are row-major bf16 with a 168-element (336 B) row stride: ds_read_b128 operand reads are conflict-free
// (20 r mod 64 covers 16 distinct 4-bank slots).  gfx950 only.
#include "common.h"
#include "../../include/ader_hip.h"

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define HP 160
#define LDR 168                 // bf16 elements per LDS / rep_bf row
#define LOG2E 1.4426950408889634f
#define RESCALE_THR 6.0f        // lazy online-softmax rescale threshold (log2 units): p <= 2^6

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// accumulator row of register `reg` for lane half hh (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// 4(k) x 16(n) transposed LDS read: lane (q = (lane&15)>>2, p = lane&3) supplies the address of row k0+q, cols n0+4p..;
// lane i of the 16-lane group receives column n0+i of rows k0..k0+3.
__device__ __forceinline__ bf16x4 tr_read(const bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)v[8 * s + j];
    return o;
}

// rep fp32 [B,H] -> rep_bf [Bp, LDR] bf16, zero padded (rows >= B, cols >= H)
__global__ __launch_bounds__(256) void k_lbf_prep(const float* __restrict__ rep, bf16* __restrict__ rep_bf, int B, int Bp, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp * LDR) return;
    const int b = i / LDR, c = i - b * LDR;
    rep_bf[i] = (bf16)((b < B && c < H) ? rep[(size_t)b * H + c] : 0.0f);
}

// bf16 shadow of the fp32 master table: shadow[row][0:H] = bf16(emb[row][0:H]), row stride LDR (336 B), padding zero.
// The logit GEMMs stream this copy (16-B pieces, LDS image == memory image); ader_adam_step keeps it in sync.
__global__ __launch_bounds__(256) void k_lbf_shadow(const float* __restrict__ emb, bf16* __restrict__ shadow, size_t rows, int H) {
    const size_t n = rows * LDR;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / LDR;
        const int c = (int)(i - r * LDR);
        shadow[i] = (bf16)((c < H) ? emb[r * H + c] : 0.0f);
    }
}

struct LbfArgs {
    int tile_off;               // first 128-item tile handled by this launch (row-sharded table update)
    const bf16* sh1;            // bf16 shadow of the table, row of item 1: rows of LDR elements (336 B), cols >= H zero
    int vrows;                  // shadow rows available from sh1 (= item_num)
    const bf16* rep_bf;         // [Bp][LDR]
    int B, Bp, H, N, ranges;
    float* pm; float* pl; float* pO;    // [ranges][Bp], [ranges][Bp], [ranges][Bp][HP]
    const float* off;           // [Bp] log2(w_b) - lse2_b
    float* demb1;               // gradient row of item 1
};

#define FB 32                      // items per streamed block
#define PCS_ROW (LDR * 2 / 16)     // 16-byte pieces per shadow row (21)
#define PCS_BLK (FB * PCS_ROW)     // pieces per block (672)
#define PPT 3                      // pieces per thread per block (256*3 >= 672)
#define RD 3                       // register ring depth: table blocks in flight per workgroup

__global__ __launch_bounds__(256, 2) void k_lbf_fwd(LbfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* E_l = (bf16*)smem_raw;                       // [2][FB][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nchunk = a.Bp >> 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range = xcd + 8 * (slot / nchunk), bc = slot % nchunk;
    if (range >= a.ranges) return;
    const int N = a.N;
    const int nblk = (N + FB - 1) / FB;
    const int per = (nblk + a.ranges - 1) / a.ranges;
    const int blk_begin = range * per, blk_end = min(nblk, blk_begin + per);
    const int nb_blocks = max(0, blk_end - blk_begin);
    const int b0 = bc * 128 + wave * 32;
    bf16x8 bfrag[10];
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) bfrag[ks] = *(const bf16x8*)(a.rep_bf + (size_t)(b0 + r) * LDR + 16 * ks + 8 * hh);
    f32x16 O[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    uint4 ring[RD][PPT];
#define LBF_LOAD(slot_, blk_)                                                                            \
    {                                                                                                    \
        const uint4* src_ = (const uint4*)(a.sh1 + (size_t)(blk_) * FB * LDR);                           \
        _Pragma("unroll") for (int j = 0; j < PPT; ++j) {                                                \
            const int idx_ = tid + 256 * j;                                                              \
            uint4 v_ = make_uint4(0u, 0u, 0u, 0u);                                                       \
            if (idx_ < PCS_BLK && (blk_) * FB + idx_ / PCS_ROW < a.vrows) v_ = src_[idx_];               \
            ring[slot_][j] = v_;                                                                         \
        }                                                                                                \
    }
#define LBF_STORE(slot_, buf_)                                                                           \
    {                                                                                                    \
        uint4* dst_ = (uint4*)(E_l + (buf_) * FB * LDR);                                                 \
        _Pragma("unroll") for (int j = 0; j < PPT; ++j) {                                                \
            const int idx_ = tid + 256 * j;                                                              \
            if (idx_ < PCS_BLK) dst_[idx_] = ring[slot_][j];                                             \
        }                                                                                                \
    }
#pragma unroll
    for (int s_ = 0; s_ < RD; ++s_) if (s_ < nb_blocks) LBF_LOAD(s_, blk_begin + s_);
    int cur = 0, i = 0;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    while (i < nb_blocks) {
#pragma unroll
        for (int s_ = 0; s_ < RD; ++s_) {
            if (i >= nb_blocks) break;                  // workgroup-uniform
            const int blk = blk_begin + i;
            LBF_STORE(s_, cur);
            if (i + RD < nb_blocks) LBF_LOAD(s_, blk + RD);
            __syncthreads();
            const bf16* Eb = E_l + cur * FB * LDR;
            const int i0 = blk * FB;
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) S[j] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                const bf16x8 af = *(const bf16x8*)(Eb + r * LDR + 16 * ks + 8 * hh);
                S = mfma_bf16(af, bfrag[ks], S);
            }
            if (i0 + FB > N) {                          // tail block: items >= N are outside the softmax
#pragma unroll
                for (int j = 0; j < 16; ++j) if (i0 + acc_row(j, hh) >= N) S[j] = -INFINITY;
            }
            float tmax = S[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, S[j]);
            // the two half-waves of a lane pair (l, l+32) hold different items of the SAME batch row; m_run is kept equal in
            // both, so the cross-half exchange is only needed on the (rare) rescale path
            float t2 = tmax * LOG2E;
            if (__any(t2 > m_run + RESCALE_THR)) {
                t2 = fmaxf(t2, __shfl_xor(t2, 32, 64));
                const float m_new = (t2 > m_run + RESCALE_THR) ? t2 : m_run;
                const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float ar = __shfl(alpha, acc_row(j, hh), 64);     // O rows are batch rows
#pragma unroll
                    for (int nb = 0; nb < 5; ++nb) O[nb][j] *= ar;
                }
            }
            const float nm = -m_run;
            float ls = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { S[j] = __builtin_amdgcn_exp2f(fmaf(S[j], LOG2E, nm)); ls += S[j]; }
            l_run += ls;
            const bf16x8 pa0 = pack8(S, 0), pa1 = pack8(S, 1);
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const bf16* base = Eb + (4 * hh + q4) * LDR + 32 * nb + 16 * g1 + 4 * p4;
                const bf16x4 l0 = tr_read(base), h0 = tr_read(base + 8 * LDR);
                const bf16x4 l1 = tr_read(base + 16 * LDR), h1 = tr_read(base + 24 * LDR);
                bf16x8 b0v, b1v;
#pragma unroll
                for (int j = 0; j < 4; ++j) { b0v[j] = l0[j]; b0v[4 + j] = h0[j]; b1v[j] = l1[j]; b1v[4 + j] = h1[j]; }
                O[nb] = mfma_bf16(pa0, b0v, O[nb]);
                O[nb] = mfma_bf16(pa1, b1v, O[nb]);
            }
            cur ^= 1;
            ++i;
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (hh == 0) {
        a.pm[(size_t)range * a.Bp + b0 + r] = m_run;
        a.pl[(size_t)range * a.Bp + b0 + r] = l_tot;
    }
    float* o = a.pO + ((size_t)range * a.Bp + b0) * HP;
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[(size_t)acc_row(j, hh) * HP + 32 * nb + r] = O[nb][j];
}

// One workgroup per batch row: merge range partials -> lse (natural log), loss row, dRep row, backward offset.
// 640 threads: thread (g = tid/160, h = tid%160) sums ranges i = g mod 4 of channel h (fixed order: deterministic).
__global__ __launch_bounds__(640) void k_lbf_combine(LbfArgs a, const int* __restrict__ lab, const float* __restrict__ wrow,
                                                     float* __restrict__ lse, float* __restrict__ off, float* __restrict__ rowloss,
                                                     float* __restrict__ drep) {
    __shared__ float sc[1024];          // per-range scale 2^(pm - M) (0 for empty ranges)
    __shared__ float red[640];
    __shared__ float sM, sL;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int R = a.ranges, H = a.H;
    if (b >= a.B) {                                  // padding rows: no loss, no gradient
        if (tid == 0) { lse[b] = 0.0f; rowloss[b] = 0.0f; off[b] = -INFINITY; }
        return;
    }
    float m = -INFINITY;
    for (int i = tid; i < R; i += 640) m = fmaxf(m, a.pm[(size_t)i * a.Bp + b]);
    red[tid] = m;
    __syncthreads();
    if (tid < 64) {
        float v = red[tid];
        for (int k = tid + 64; k < 640; k += 64) v = fmaxf(v, red[k]);
        v = wave_max(v);
        if (tid == 0) sM = v;
    }
    __syncthreads();
    const float M = sM;
    float l = 0.0f;
    for (int i = tid; i < R; i += 640) {
        const float pm = a.pm[(size_t)i * a.Bp + b];
        const float s_ = (pm != -INFINITY) ? __builtin_amdgcn_exp2f(pm - M) : 0.0f;
        sc[i] = s_;
        l += a.pl[(size_t)i * a.Bp + b] * s_;
    }
    red[tid] = l;
    __syncthreads();
    if (tid == 0) {                                  // fixed order
        float v = 0.0f;
        for (int k = 0; k < 640; ++k) v += red[k];
        sL = v;
    }
    __syncthreads();
    const float L = sL;
    const float lse2 = M + log2f(L);
    const int t = lab[b] - 1;
    const float w = wrow[b];
    const int g = tid / 160, h = tid - g * 160;
    float oh = 0.0f;
    if (h < H) {
#pragma unroll 8
        for (int i = g; i < R; i += 4) oh += a.pO[((size_t)i * a.Bp + b) * HP + h] * sc[i];
    }
    __syncthreads();
    red[tid] = oh;
    __syncthreads();
    float part = 0.0f, et = 0.0f;
    if (tid < H) {
        oh = ((red[tid] + red[160 + tid]) + red[320 + tid]) + red[480 + tid];
        if (t >= 0) {                                // target logit with the same bf16-rounded operands as the MFMA path
            et = (float)a.sh1[(size_t)t * LDR + tid];
            part = (float)a.rep_bf[(size_t)b * LDR + tid] * et;
        }
        drep[(size_t)b * H + tid] = w * (oh / L - et);
    }
    __syncthreads();
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
        float s_lab = 0.0f;
        for (int k = 0; k < H; ++k) s_lab += red[k];
        const float z = lse2 / LOG2E;
        lse[b] = z;
        rowloss[b] = (t >= 0) ? w * (z - s_lab) : 0.0f;
        off[b] = (w > 0.0f) ? log2f(w) - lse2 : -INFINITY;
    }
}

// Catalog-sharded data parallelism: this rank streamed only ITS items.  Per batch row, merge the range partials into one
// (M, L, O[H]) triple -- running max (log2 domain), sum of 2^(s - M), sum of 2^(s - M) * E -- for the cross-rank merge
// (same fixed-order sums as k_lbf_combine).  part: [Bp][PART_LD] = {M, L, O[0..H)}.
#define PART_LD 152
__global__ __launch_bounds__(640) void k_lbf_combine_partial(LbfArgs a, float* __restrict__ part) {
    __shared__ float sc[1024];
    __shared__ float red[640];
    __shared__ float sM;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int R = a.ranges, H = a.H;
    float m = -INFINITY;
    for (int i = tid; i < R; i += 640) m = fmaxf(m, a.pm[(size_t)i * a.Bp + b]);
    red[tid] = m;
    __syncthreads();
    if (tid < 64) {
        float v = red[tid];
        for (int k = tid + 64; k < 640; k += 64) v = fmaxf(v, red[k]);
        v = wave_max(v);
        if (tid == 0) sM = v;
    }
    __syncthreads();
    const float M = sM;
    float l = 0.0f;
    for (int i = tid; i < R; i += 640) {
        const float pm = a.pm[(size_t)i * a.Bp + b];
        const float s_ = (pm != -INFINITY) ? __builtin_amdgcn_exp2f(pm - M) : 0.0f;
        sc[i] = s_;
        l += a.pl[(size_t)i * a.Bp + b] * s_;
    }
    red[tid] = l;
    __syncthreads();
    if (tid == 0) {
        float v = 0.0f;
        for (int k = 0; k < 640; ++k) v += red[k];
        part[(size_t)b * PART_LD + 0] = M;
        part[(size_t)b * PART_LD + 1] = v;
    }
    const int g = tid / 160, h = tid - g * 160;
    float oh = 0.0f;
    if (h < H) {
#pragma unroll 8
        for (int i = g; i < R; i += 4) oh += a.pO[((size_t)i * a.Bp + b) * HP + h] * sc[i];
    }
    __syncthreads();
    red[tid] = oh;
    __syncthreads();
    if (tid < H) part[(size_t)b * PART_LD + 2 + tid] = ((red[tid] + red[160 + tid]) + red[320 + tid]) + red[480 + tid];
}

// Cross-rank merge for one batch row of THIS rank: parts [W][Bp][PART_LD] (slice i = the partials rank i computed over its
// items) -> lse (natural log), backward offset, loss row, dRep row.  e_lab: fp32 table row of the label (fetched from its
// owner); it enters as bf16(e_lab), i.e. exactly the shadow row the MFMA path multiplies.  One wave per row.
__global__ __launch_bounds__(256) void k_lbf_merge_parts(const float* __restrict__ parts, int W, int Bp, int B, int H,
                                                         const float* __restrict__ e_lab, const bf16* __restrict__ rep_bf,
                                                         const float* __restrict__ wrow, float* __restrict__ lse,
                                                         float* __restrict__ off, float* __restrict__ rowloss,
                                                         float* __restrict__ drep) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= Bp) return;
    if (b >= B) {
        if (lane == 0) { lse[b] = 0.0f; rowloss[b] = 0.0f; off[b] = -INFINITY; }
        return;
    }
    float M = -INFINITY;
    for (int i = 0; i < W; ++i) M = fmaxf(M, parts[((size_t)i * Bp + b) * PART_LD]);
    float L = 0.0f, o[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < W; ++i) {                                   // fixed order: deterministic
        const float* pr = parts + ((size_t)i * Bp + b) * PART_LD;
        const float m = pr[0];
        const float sc = (m != -INFINITY) ? __builtin_amdgcn_exp2f(m - M) : 0.0f;
        L += pr[1] * sc;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int c = lane + 64 * k; if (c < H) o[k] += pr[2 + c] * sc; }
    }
    const float lse2 = M + log2f(L);
    const float w = wrow[b];
    float dot = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c = lane + 64 * k;
        if (c < H) {
            const float et = (float)(bf16)e_lab[(size_t)b * H + c];
            dot += (float)rep_bf[(size_t)b * LDR + c] * et;
            drep[(size_t)b * H + c] = w * (o[k] / L - et);
        }
    }
    dot = wave_sum(dot);
    if (lane == 0) {
        const float z = lse2 / LOG2E;
        lse[b] = z;
        rowloss[b] = w * (z - dot);
        off[b] = (w > 0.0f) ? log2f(w) - lse2 : -INFINITY;
    }
}

__global__ __launch_bounds__(256) void k_lbf_sum(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) acc += x[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

// Optional fused optimiser ("the table gradient is never materialised"): when the whole gradient of a table row is
// available inside the workgroup that owns it -- the dense logits term from the MFMAs, plus the sparse input-embedding
// rows and one-hot target rows looked up in id-sorted lists -- the TF-Adam update of that row (ADER.py:96) is applied in
// place from the LDS staging tile: theta/m/v are read and written once, dE is neither written nor re-read, and the
// bf16 shadow row is refreshed on the way out.  Sparse terms are added in list order (deterministic, no atomics).
struct FuseArgs {
    const int* sp_ids; const int* sp_rows; int n_sp; const float* sp_src; float sp_scale;   // input-embedding rows (sorted by id)
    const int* tg_ids; const int* tg_rows; int n_tg; const float* wrow;                      // one-hot targets (sorted by id)
    const int* sp_start; const int* tg_start;   // bucket offsets into the two lists: bucket j = ids [gran*j + id0, gran*(j+1) + id0)
    float* emb1; float* m1; float* v1; bf16* sh1w;                                           // row of item 1 of theta/m/v/shadow
    float lr_t, omb1, omb2, eps;
    const float* extra1;        // EXTRA: dense gradient rows to add (row of item 1; [.,H] fp32), e.g. distilled rows' term
};

__device__ __forceinline__ int lower_bound_i32(const int* __restrict__ a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// dE tile (128 items per workgroup, 32 per wave); loops over all batch rows in chunks of 64 staged through LDS.
#define FLD 152                    // fp32 row stride of the dE staging tile
#ifndef NT_STORES
#define NT_STORES 1
#endif

// workgroup barrier that orders LDS traffic only (__syncthreads also drains every outstanding global access of the wave)
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool ADAM, bool EXTRA = false>
__global__ __launch_bounds__(256, 2) void k_lbf_bwd_de(LbfArgs a, FuseArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R_l = (bf16*)smem_raw;                        // [2][64][LDR]  (also: the table tile, then the dE staging tile)
    float* off_l = (float*)(smem_raw + 2 * 64 * LDR * sizeof(bf16));   // [Bp]
    int* meta_l = (int*)(off_l + a.Bp);      // ADAM: per (half, list): [k0, k1, 8 x (id, row)] = 18 ints, 4 lists (SP_PRE entries prefetched)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int H = a.H, N = a.N;
    const int tile0 = (blockIdx.x + a.tile_off) * 128;
    const int it0 = tile0 + wave * 32;
    {   // table tile: 128 shadow rows, contiguous -> LDS (coalesced 16-B pieces) -> operand fragments in registers
        const uint4* src = (const uint4*)(a.sh1 + (size_t)tile0 * LDR);
        uint4* dst = (uint4*)R_l;
        for (int idx = tid; idx < 128 * PCS_ROW; idx += 256) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (tile0 + idx / PCS_ROW < a.vrows) v = src[idx];
            dst[idx] = v;
        }
    }
    for (int i = tid; i < a.Bp; i += 256) off_l[i] = a.off[i];
    if (ADAM && tid < 4) {
        // the sparse lists of the two half-tiles (bucket bounds and the first entries) are fetched now, under the GEMM phase:
        // three dependent global round trips less between the GEMM and the streaming update
        const int half = tid >> 1, lst = tid & 1;
        const int bkt = (tile0 + half * 64) >> 6;
        const int* st = lst ? f.tg_start : f.sp_start;
        const int* ids = lst ? f.tg_ids : f.sp_ids;
        const int* rows = lst ? f.tg_rows : f.sp_rows;
        int* mt = meta_l + tid * 18;
        int k0 = 0, k1 = 0;
        if (tile0 + half * 64 < N) { k0 = st[bkt]; k1 = st[bkt + 1]; }
        mt[0] = k0; mt[1] = k1;
        for (int i = 0; i < 8 && k0 + i < k1; ++i) { mt[2 + 2 * i] = ids[k0 + i]; mt[3 + 2 * i] = rows[k0 + i]; }
    }
    __syncthreads();
    // ADAM: the first SPV input-embedding gradient rows of each half-tile (thread c holds column c), requested now and
    // consumed after the GEMM phase
#define SPV 3
    float spv0[SPV], spv1[SPV];
    if (ADAM) {
#pragma unroll
        for (int i = 0; i < SPV; ++i) {
            const int* m0 = meta_l, * m1 = meta_l + 2 * 18;
            spv0[i] = (tid < H && m0[0] + i < m0[1]) ? f.sp_src[(size_t)m0[3 + 2 * i] * H + tid] * f.sp_scale : 0.0f;
            spv1[i] = (tid < H && m1[0] + i < m1[1]) ? f.sp_src[(size_t)m1[3 + 2 * i] * H + tid] * f.sp_scale : 0.0f;
        }
    }
    bf16x8 efrag[10];                                   // lane (item r, half hh) holds E[item][16ks + 8hh + 0..7]
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) efrag[ks] = *(const bf16x8*)(R_l + (wave * 32 + r) * LDR + 16 * ks + 8 * hh);
    __syncthreads();
    f32x16 dE[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) dE[nb][j] = 0.0f;
    const int nch = a.Bp >> 6;
    const int n16 = 64 * LDR * 2 / 16;                  // 16-byte pieces per 64-row chunk (1344)
    uint4 pf[6];
#define LBF_RPREFETCH(c_)                                                                               \
    {                                                                                                   \
        const uint4* src_ = (const uint4*)(a.rep_bf + (size_t)(c_) * 64 * LDR);                          \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                 \
            const int idx = tid + 256 * j;                                                              \
            pf[j] = (idx < n16) ? src_[idx] : make_uint4(0u, 0u, 0u, 0u);                               \
        }                                                                                               \
    }
#define LBF_RSTAGE(buf_)                                                                                \
    {                                                                                                   \
        uint4* dst_ = (uint4*)(R_l + (buf_) * 64 * LDR);                                                 \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                 \
            const int idx = tid + 256 * j;                                                              \
            if (idx < n16) dst_[idx] = pf[j];                                                           \
        }                                                                                               \
    }
    LBF_RPREFETCH(0); LBF_RSTAGE(0);
    __syncthreads();
    int cur = 0;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    for (int c = 0; c < nch; ++c) {
        const bool more = c + 1 < nch;
        if (more) LBF_RPREFETCH(c + 1);
        const bf16* Rb = R_l + cur * 64 * LDR;
#pragma unroll 1
        for (int bb = 0; bb < 2; ++bb) {
            const int b0 = c * 64 + bb * 32;
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) S[j] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                const bf16x8 af = *(const bf16x8*)(Rb + (bb * 32 + r) * LDR + 16 * ks + 8 * hh);
                S = mfma_bf16(af, efrag[ks], S);
            }
            // rows of S are batch rows: p = w_b * softmax = exp2(S*log2e + off_b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 o4 = *(const float4*)(off_l + b0 + 8 * g + 4 * hh);
                S[4 * g + 0] = __builtin_amdgcn_exp2f(fmaf(S[4 * g + 0], LOG2E, o4.x));
                S[4 * g + 1] = __builtin_amdgcn_exp2f(fmaf(S[4 * g + 1], LOG2E, o4.y));
                S[4 * g + 2] = __builtin_amdgcn_exp2f(fmaf(S[4 * g + 2], LOG2E, o4.z));
                S[4 * g + 3] = __builtin_amdgcn_exp2f(fmaf(S[4 * g + 3], LOG2E, o4.w));
            }
            const bf16x8 pa0 = pack8(S, 0), pa1 = pack8(S, 1);
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const bf16* base = Rb + (bb * 32 + 4 * hh + q4) * LDR + 32 * nb + 16 * g1 + 4 * p4;
                const bf16x4 l0 = tr_read(base), h0 = tr_read(base + 8 * LDR);
                const bf16x4 l1 = tr_read(base + 16 * LDR), h1 = tr_read(base + 24 * LDR);
                bf16x8 b0v, b1v;
#pragma unroll
                for (int j = 0; j < 4; ++j) { b0v[j] = l0[j]; b0v[4 + j] = h0[j]; b1v[j] = l1[j]; b1v[4 + j] = h1[j]; }
                dE[nb] = mfma_bf16(pa0, b0v, dE[nb]);
                dE[nb] = mfma_bf16(pa1, b1v, dE[nb]);
            }
        }
        if (more) LBF_RSTAGE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // dE acc (rows = items, col = channel) -> LDS [64 items][fs] -> coalesced row stores, two halves of 64 items.
    // ADAM: fs = H, so the LDS tile is the same flat [64*H] block as the half-tile's rows of theta / m / v in memory.
    float* F_l = (float*)smem_raw;
    const int HH = H >> 1;
    const int fs = ADAM ? H : FLD;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define AV 6
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        // ADAM: the first round of theta/m/v vectors of this half-tile is requested BEFORE the dE staging and the sparse
        // terms (independent of both); the barriers in between order LDS only, so the loads stay in flight across them
        const int base_it = tile0 + half * 64;
        const int rows_valid = min(64, N - base_it);
        const int n_el = rows_valid > 0 ? rows_valid * H : 0;
        float* __restrict__ gp = ADAM ? f.emb1 + (size_t)base_it * H : nullptr;
        float* __restrict__ gm = ADAM ? f.m1 + (size_t)base_it * H : nullptr;
        float* __restrict__ gv = ADAM ? f.v1 + (size_t)base_it * H : nullptr;
        const int head = (((uintptr_t)gp) & 15) ? 2 : 0;
        int e = head + 4 * tid;
        int row = e / H, col = e - row * H;
        const int step_r = 1024 / H, step_c = 1024 - step_r * H;
        f32x4_t P[AV], M[AV], V[AV], G[EXTRA ? AV : 1];
        const float* __restrict__ gx = EXTRA ? f.extra1 + (size_t)base_it * H : nullptr;
        int E[AV], RC[AV], NV[AV];
#define ROUND_LOAD()                                                                                       \
        _Pragma("unroll") for (int u = 0; u < AV; ++u) {                                                   \
            E[u] = e; RC[u] = (row << 16) | col;                                                           \
            NV[u] = (e + 3 < n_el) ? 2 : ((e + 1 < n_el) ? 1 : 0);                                         \
            if (NV[u] == 2) {                                                                              \
                P[u] = *(const f32x4_t*)(gp + e); M[u] = *(const f32x4_t*)(gm + e); V[u] = *(const f32x4_t*)(gv + e); \
                if (EXTRA) G[u] = __builtin_nontemporal_load((const f32x4_t*)(gx + e));                    \
            } else if (NV[u] == 1) {                                                                       \
                if (EXTRA) { const f32x2_t g_ = *(const f32x2_t*)(gx + e); G[u] = (f32x4_t){g_[0], g_[1], 0.f, 0.f}; } \
                const f32x2_t p = *(const f32x2_t*)(gp + e), m = *(const f32x2_t*)(gm + e), v = *(const f32x2_t*)(gv + e); \
                P[u] = (f32x4_t){p[0], p[1], 0.f, 0.f}; M[u] = (f32x4_t){m[0], m[1], 0.f, 0.f}; V[u] = (f32x4_t){v[0], v[1], 0.f, 0.f}; \
            }                                                                                              \
            e += 1024; row += step_r; col += step_c;                                                       \
            if (col >= H) { col -= H; ++row; }                                                             \
        }
        if (ADAM) { ROUND_LOAD(); }
        if (ADAM) lds_only_barrier(); else __syncthreads();
        if ((wave >> 1) == half) {
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const int h = 32 * nb + r;
                if (h < fs) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) F_l[((wave & 1) * 32 + acc_row(j, hh)) * fs + h] = dE[nb][j];
                }
            }
        }
        if (ADAM) lds_only_barrier(); else __syncthreads();
        if (!ADAM) {
            for (int idx = tid; idx < 64 * HH; idx += 256) {
                const int row = idx / HH, c2 = idx - row * HH;
                if (base_it + row < N)
                    *(float2*)(a.demb1 + (size_t)(base_it + row) * H + 2 * c2) = *(const float2*)(F_l + row * FLD + 2 * c2);
            }
        } else {
            // sparse terms of this half-tile: item ids [base_it+1, base_it+65).  Thread c owns column c of every row.
            const int id_lo = base_it + 1, id_hi = min(base_it + 64, N) + 1;
            if (tid < H && id_lo < id_hi) {
                // entries of bucket base_it >> 6 (ids [base_it+1, base_it+65): exactly this half-tile), (id, row)-ordered
                const int* ms = meta_l + (half * 2 + 0) * 18;
                const int* mg = meta_l + (half * 2 + 1) * 18;
                const int k0s = ms[0], k1s = ms[1];
#pragma unroll
                for (int i = 0; i < SPV; ++i) {                  // rows already in registers (same (id, row) order)
                    if (k0s + i < k1s) {
                        const int id = ms[2 + 2 * i];
                        if (id < id_hi) F_l[(id - id_lo) * fs + tid] += half ? spv1[i] : spv0[i];
                    }
                }
                for (int k = k0s + SPV, i = SPV; k < k1s; ++k, ++i) {
                    const int id = (i < 8) ? ms[2 + 2 * i] : f.sp_ids[k];
                    if (id >= id_hi) break;
                    const int row = (i < 8) ? ms[3 + 2 * i] : f.sp_rows[k];
                    F_l[(id - id_lo) * fs + tid] += f.sp_src[(size_t)row * H + tid] * f.sp_scale;
                }
                for (int k = mg[0], k1 = mg[1], i = 0; k < k1; ++k, ++i) {
                    const int id = (i < 8) ? mg[2 + 2 * i] : f.tg_ids[k];
                    if (id >= id_hi) break;
                    const int b = (i < 8) ? mg[3 + 2 * i] : f.tg_rows[k];
                    F_l[(id - id_lo) * fs + tid] -= f.wrow[b] * (float)a.rep_bf[(size_t)b * LDR + tid];
                }
            }
            lds_only_barrier();
            // Adam on the half-tile.  Its rows are ONE contiguous block of 64*H floats in theta / m / v (and in F_l): it is
            // walked as 16-byte vectors (the block starts 0 or 8 bytes past a 16-byte boundary: `head` floats are peeled),
            // all loads of a round issued before any math or store; (row, col) of a vector -- needed only for the bf16
            // shadow row -- is stepped without divisions.
            bf16* __restrict__ psh = f.sh1w + (size_t)base_it * LDR;
#define ADAM1(p_, m_, v_, g_)                                                                              \
            { m_ += ((g_) - m_) * f.omb1; v_ += ((g_) * (g_) - v_) * f.omb2; p_ -= (m_ * f.lr_t) / (sqrtf(v_) + f.eps); }
            if (head && tid == 0 && n_el > 0) {                   // elements 0,1 (row 0, columns 0,1)
                f32x2_t p = *(const f32x2_t*)gp, m = *(const f32x2_t*)gm, v = *(const f32x2_t*)gv;
                float2 g2 = *(const float2*)F_l;
                if (EXTRA) { g2.x += gx[0]; g2.y += gx[1]; }
                ADAM1(p[0], m[0], v[0], g2.x); ADAM1(p[1], m[1], v[1], g2.y);
                *(f32x2_t*)gp = p; *(f32x2_t*)gm = m; *(f32x2_t*)gv = v;
                bf16x2 sb; sb[0] = (bf16)p[0]; sb[1] = (bf16)p[1];
                *(bf16x2*)psh = sb;
            }
#pragma unroll 1
            for (int k0 = 0; k0 < 12; k0 += AV) {                 // 12 * 1024 floats >= 64 * 160; round 0 is already in flight
                if (k0) { ROUND_LOAD(); }
#pragma unroll
                for (int u = 0; u < AV; ++u) {
                    if (NV[u] == 0) continue;
                    float2 ga = *(const float2*)(F_l + E[u]);
                    float2 gb = (NV[u] == 2) ? *(const float2*)(F_l + E[u] + 2) : make_float2(0.f, 0.f);
                    if (EXTRA) { ga.x += G[u][0]; ga.y += G[u][1]; gb.x += G[u][2]; gb.y += G[u][3]; }
                    f32x4_t p = P[u], m = M[u], v = V[u];
                    ADAM1(p[0], m[0], v[0], ga.x); ADAM1(p[1], m[1], v[1], ga.y);
                    ADAM1(p[2], m[2], v[2], gb.x); ADAM1(p[3], m[3], v[3], gb.y);
                    const int r0 = RC[u] >> 16, c0 = RC[u] & 0xffff;
                    bf16x2 s0; s0[0] = (bf16)p[0]; s0[1] = (bf16)p[1];
                    *(bf16x2*)(psh + r0 * LDR + c0) = s0;
                    if (NV[u] == 2) {
                        // theta/m/v of this block are not touched again this step: keep them out of the caches
                        __builtin_nontemporal_store(p, (f32x4_t*)(gp + E[u]));
                        __builtin_nontemporal_store(m, (f32x4_t*)(gm + E[u]));
                        __builtin_nontemporal_store(v, (f32x4_t*)(gv + E[u]));
                        const int c1 = c0 + 2;
                        bf16x2 s1; s1[0] = (bf16)p[2]; s1[1] = (bf16)p[3];
                        *(bf16x2*)(psh + ((c1 >= H) ? (r0 + 1) * LDR + (c1 - H) : r0 * LDR + c1)) = s1;
                    } else {
                        *(f32x2_t*)(gp + E[u]) = (f32x2_t){p[0], p[1]};
                        *(f32x2_t*)(gm + E[u]) = (f32x2_t){m[0], m[1]};
                        *(f32x2_t*)(gv + E[u]) = (f32x2_t){v[0], v[1]};
                    }
                }
            }
#undef ADAM1
        }
    }
}

// sparse one-hot term of dlogit: dE[label_b,:] -= w_b * rep_b  (one wave per batch row, float atomics)
__global__ __launch_bounds__(256) void k_lbf_target_fix(const bf16* __restrict__ rep_bf, const int* __restrict__ lab,
                                                        const float* __restrict__ wrow, float* __restrict__ demb1, int B, int H) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int t = lab[b] - 1;
    if (t < 0) return;
    const float w = wrow[b];
    for (int c = lane; c < H; c += 64) atomicAdd(demb1 + (size_t)t * H + c, -w * (float)rep_bf[(size_t)b * LDR + c]);
}

// ============================================================================================= C ABI
static const size_t kFwdLds = (size_t)2 * FB * LDR * sizeof(bf16);
static size_t bwd_lds(int Bp) { return (size_t)2 * 64 * LDR * sizeof(bf16) + (size_t)Bp * sizeof(float) + 4 * 18 * sizeof(int); }

extern "C" {

// shadow [rows][168] bf16 <- emb [rows][H] fp32  (initialisation / checkpoint load; ader_adam_step keeps it in sync afterwards)
int ader_lbf_shadow_refresh(const float* emb, void* shadow, size_t rows, int H, void* stream) {
    if (rows == 0) return 0;
    if (H > HP) return -2;
    size_t g = (rows * LDR + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_lbf_shadow, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, emb, (bf16*)shadow, rows, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_lbf_ranges(int N, int Bp) {
    const int nblk = (N + FB - 1) / FB;
    const int nchunk = Bp / 128;
    int target = 512 / (nchunk < 1 ? 1 : nchunk);
    int r = 8;
    while (r * 2 <= target && r * 2 <= nblk) r *= 2;
    return r;
}

// Forward of the one-hot softmax CE over items 1..N with bf16 MFMA.  Bp % 128 == 0, H even, H <= 160.
// Scratch: rep_bf Bp*168 bf16; pm, pl: ranges*Bp floats; pO: ranges*Bp*160 floats (ranges = ader_lbf_ranges(N,Bp)).
// Outputs: lse [Bp] (natural log), off [Bp] (backward exponent offsets), rowloss [Bp], loss [1], drep [B,H].
int ader_lbf_fwd(const float* rep, const void* shadow, int item_num, int B, int Bp, int H, int N, const int* lab, const float* wrow,
                 void* rep_bf, float* pm, float* pl, float* pO, float* lse, float* off, float* rowloss, float* loss, float* drep,
                 void* stream) {
    if (B <= 0) return 0;
    if (Bp % 128 != 0 || B > Bp || H > HP || (H & 1) || H < 2) return -2;
    static bool f = false;
    if (!f) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lbf_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds);
        if (e != hipSuccess) return (int)e;
        f = true;
    }
    hipStream_t st = (hipStream_t)stream;
    LbfArgs a;
    if (N > item_num) return -2;
    a.sh1 = (const bf16*)shadow + LDR; a.vrows = item_num; a.tile_off = 0;
    a.rep_bf = (const bf16*)rep_bf; a.B = B; a.Bp = Bp; a.H = H; a.N = N; a.ranges = ader_lbf_ranges(N, Bp);
    a.pm = pm; a.pl = pl; a.pO = pO; a.off = nullptr; a.demb1 = nullptr;
    hipLaunchKernelGGL(k_lbf_prep, dim3((Bp * LDR + 255) / 256), dim3(256), 0, st, rep, (bf16*)rep_bf, B, Bp, H);
    hipLaunchKernelGGL(k_lbf_fwd, dim3(a.ranges * (Bp / 128)), dim3(256), kFwdLds, st, a);
    hipLaunchKernelGGL(k_lbf_combine, dim3(Bp), dim3(640), 0, st, a, lab, wrow, lse, off, rowloss, drep);
    hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, st, rowloss, B, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}

// rep fp32 [B,H] -> rep_bf [Bp,168] bf16 (zero padded): the operand layout of the bf16 logit kernels
int ader_lbf_prep(const float* rep, void* rep_bf, int B, int Bp, int H, void* stream) {
    if (Bp <= 0) return 0;
    if (B > Bp || H > HP) return -2;
    hipLaunchKernelGGL(k_lbf_prep, dim3((Bp * LDR + 255) / 256), dim3(256), 0, (hipStream_t)stream, rep, (bf16*)rep_bf, B, Bp, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Softmax partials of ALL Bp batch rows over the item shard [item_begin+1, item_begin+item_count] (clipped to N): the
// forward of a catalog-sharded rank.  part [Bp][152] = {M (log2 domain), L, O[0..H)} per row; rows whose shard is empty get
// {-inf, 0, 0}.  Scratch pm/pl/pO sized with ader_lbf_ranges(item_count, Bp).
int ader_lbf_fwd_shard(const void* rep_bf, const void* shadow, int item_num, int Bp, int H, int N, int item_begin, int item_count,
                       float* pm, float* pl, float* pO, float* part, void* stream) {
    if (Bp <= 0) return 0;
    if (Bp % 128 != 0 || H > HP || (H & 1) || H < 2 || N > item_num || item_begin < 0) return -2;
    static bool f = false;
    if (!f) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lbf_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds);
        if (e != hipSuccess) return (int)e;
        f = true;
    }
    hipStream_t st = (hipStream_t)stream;
    int n_loc = N - item_begin;
    if (n_loc > item_count) n_loc = item_count;
    if (n_loc < 0) n_loc = 0;
    LbfArgs a;
    a.sh1 = (const bf16*)shadow + (size_t)LDR * (1 + item_begin); a.vrows = item_num - item_begin; a.tile_off = 0;
    a.rep_bf = (const bf16*)rep_bf; a.B = Bp; a.Bp = Bp; a.H = H; a.N = n_loc;
    a.ranges = n_loc > 0 ? ader_lbf_ranges(n_loc, Bp) : 0;
    a.pm = pm; a.pl = pl; a.pO = pO; a.off = nullptr; a.demb1 = nullptr;
    if (a.ranges > 0) hipLaunchKernelGGL(k_lbf_fwd, dim3(a.ranges * (Bp / 128)), dim3(256), kFwdLds, st, a);
    hipLaunchKernelGGL(k_lbf_combine_partial, dim3(Bp), dim3(640), 0, st, a, part);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Merge of the W ranks' partials (ader_lbf_fwd_shard, exchanged so that slice i holds rank i's partials of THIS rank's rows)
// into lse / off / rowloss [Bp], loss [1] and drep [B,H]; e_lab [B,H]: fp32 table rows of the labels.
int ader_lbf_merge_parts(const float* parts, int world, int Bp, int B, int H, const float* e_lab, const void* rep_bf,
                         const float* wrow, float* lse, float* off, float* rowloss, float* loss, float* drep, void* stream) {
    if (Bp <= 0) return 0;
    if (B > Bp || H > 192 || world < 1) return -2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_lbf_merge_parts, dim3((Bp + 3) / 4), dim3(256), 0, st, parts, world, Bp, B, H, e_lab, (const bf16*)rep_bf,
                       wrow, lse, off, rowloss, drep);
    hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, st, rowloss, B, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Table gradient rows 1..N (overwritten), including the sparse one-hot term.
int ader_lbf_bwd_demb(const void* rep_bf, const void* shadow, int item_num, int B, int Bp, int H, int N, const int* lab,
                      const float* wrow, const float* off, float* demb, void* stream) {
    if (B <= 0) return 0;
    if (Bp % 128 != 0 || B > Bp || H > HP || (H & 1) || H < 2) return -2;
    static bool f = false;
    static int lds_set = 0;
    const size_t lds = bwd_lds(Bp);
    if (!f || (int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lbf_bwd_de<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        f = true; lds_set = (int)lds;
    }
    hipStream_t st = (hipStream_t)stream;
    LbfArgs a;
    if (N > item_num) return -2;
    a.sh1 = (const bf16*)shadow + LDR; a.vrows = item_num; a.tile_off = 0;
    a.rep_bf = (const bf16*)rep_bf; a.B = B; a.Bp = Bp; a.H = H; a.N = N; a.ranges = 0;
    a.pm = a.pl = a.pO = nullptr; a.off = off; a.demb1 = demb + H;
    FuseArgs fa = {};
    hipLaunchKernelGGL(k_lbf_bwd_de<false>, dim3((N + 127) / 128), dim3(256), lds, st, a, fa);
    hipLaunchKernelGGL(k_lbf_target_fix, dim3((B + 3) / 4), dim3(256), 0, st, (const bf16*)rep_bf, lab, wrow, demb + H, B, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Fused: table-gradient GEMM + sparse terms + TF-Adam on table rows 1..N + shadow refresh, in one pass (single GPU).
// sp_ids/sp_rows: the B*T input positions sorted by item id (pads = id 0 first) and their row index into sp_src [B*T,H]
// (the masked/dropout-scaled gradient rows left by ader_embed_bwd_rows); sp_scale = sqrt(H).
// tg_ids/tg_rows: the B labels sorted by id and their batch row.  emb/adam_m/adam_v: fp32 [item_num+1, H].
int ader_lbf_bwd_adam_ex(const void* rep_bf, void* shadow, int item_num, int B, int Bp, int H, int N, const float* off,
                         const int* sp_ids, const int* sp_rows, const int* sp_start, int n_sp, const float* sp_src, float sp_scale,
                         const int* tg_ids, const int* tg_rows, const int* tg_start, int n_tg, const float* wrow, float* emb,
                         float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                         int tile_count, const float* extra_grad, void* stream);

int ader_lbf_bwd_adam(const void* rep_bf, void* shadow, int item_num, int B, int Bp, int H, int N, const float* off,
                      const int* sp_ids, const int* sp_rows, const int* sp_start, int n_sp, const float* sp_src, float sp_scale,
                      const int* tg_ids, const int* tg_rows, const int* tg_start, int n_tg, const float* wrow, float* emb,
                      float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                      int tile_count, void* stream) {
    return ader_lbf_bwd_adam_ex(rep_bf, shadow, item_num, B, Bp, H, N, off, sp_ids, sp_rows, sp_start, n_sp, sp_src, sp_scale, tg_ids,
                                tg_rows, tg_start, n_tg, wrow, emb, adam_m, adam_v, lr_t, beta1, beta2, eps, tile_begin, tile_count,
                                nullptr, stream);
}

// As ader_lbf_bwd_adam, plus a dense gradient extra_grad [item_num+1, H] (fp32, table layout; NULL = none) added row by row
// before the update -- the table gradient of rows that did not go through the bf16 logit path (distilled exemplar rows).
int ader_lbf_bwd_adam_ex(const void* rep_bf, void* shadow, int item_num, int B, int Bp, int H, int N, const float* off,
                         const int* sp_ids, const int* sp_rows, const int* sp_start, int n_sp, const float* sp_src, float sp_scale,
                         const int* tg_ids, const int* tg_rows, const int* tg_start, int n_tg, const float* wrow, float* emb,
                         float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                         int tile_count, const float* extra_grad, void* stream) {
    if (B <= 0) return 0;
    if (Bp % 128 != 0 || B > Bp || H > HP || (H & 1) || H < 2 || N > item_num) return -2;
    static bool f = false;
    static int lds_set = 0;
    const size_t lds = bwd_lds(Bp);
    if (!f || (int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lbf_bwd_de<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)k_lbf_bwd_de<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        f = true; lds_set = (int)lds;
    }
    LbfArgs a;
    a.sh1 = (const bf16*)shadow + LDR; a.vrows = item_num; a.tile_off = 0;
    a.rep_bf = (const bf16*)rep_bf; a.B = B; a.Bp = Bp; a.H = H; a.N = N; a.ranges = 0;
    a.pm = a.pl = a.pO = nullptr; a.off = off; a.demb1 = nullptr;
    FuseArgs fa;
    fa.sp_ids = sp_ids; fa.sp_rows = sp_rows; fa.n_sp = n_sp; fa.sp_src = sp_src; fa.sp_scale = sp_scale;
    fa.tg_ids = tg_ids; fa.tg_rows = tg_rows; fa.n_tg = n_tg; fa.wrow = wrow;
    fa.sp_start = sp_start; fa.tg_start = tg_start;
    fa.emb1 = emb + H; fa.m1 = adam_m + H; fa.v1 = adam_v + H; fa.sh1w = (bf16*)shadow + LDR;
    fa.lr_t = lr_t; fa.omb1 = 1.0f - beta1; fa.omb2 = 1.0f - beta2; fa.eps = eps;
    fa.extra1 = extra_grad ? extra_grad + H : nullptr;
    {   // tiles [tile_begin, tile_begin + tile_count) of the ceil(N/128) item tiles (tile_count < 0: all)
        const int all = (N + 127) / 128;
        int tb = tile_begin < 0 ? 0 : tile_begin;
        int te = tile_count < 0 ? all : tb + tile_count;
        if (te > all) te = all;
        if (te <= tb) return 0;
        a.tile_off = tb;
        if (extra_grad) hipLaunchKernelGGL((k_lbf_bwd_de<true, true>), dim3(te - tb), dim3(256), lds, (hipStream_t)stream, a, fa);
        else hipLaunchKernelGGL((k_lbf_bwd_de<true, false>), dim3(te - tb), dim3(256), lds, (hipStream_t)stream, a, fa);
    }
    HIP_LAUNCH_CHECK();
    return 0;
}

// bucket layout the caller must use for sp_start / tg_start: granularity (ids per bucket) and first id of bucket 0
// (a half-tile of the update covers item ids [64j + 1, 64j + 65))
int ader_fused_bucket_gran(void) { return 64; }
int ader_fused_bucket_id0(void) { return 1; }

}  // extern "C"
