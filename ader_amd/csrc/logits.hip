// Full-catalog logits, softmax cross-entropy / distillation loss and their gradients, float32 path.
// Reference: ADER.py:88-93 (logits = rep . item_emb^T over items 1..N, one-hot CE), ADER.py:108-137
// (ADER loss: CE on the train rows + lambda * soft-label CE of the exemplar rows against softmax(teacher)
// over the first Np columns), ADER.py:99-103 (rank of every item = argsort(argsort(-logits))).
//
// The [B,N] logits / one-hot / softmax tensors of the TF graph are never materialised: every pass
// recomputes 64x64 logit tiles on v_mfma_f32_16x16x4_f32 from LDS-staged table rows (each table row is
// read once per pass, coalesced) and reduces them on the fly:
//   k_logits_tile<LSE>   per-row online (max, sum-exp, target.logit) partials      -> k_lse_loss
//   k_logits_tile<RANK>  per-row count of items ranked before the target           (Evaluator, util.py:323-325)
//   k_logits_tile<STORE> dense logits (teacher logits of selected exemplars, util.py:433; model.logits fetch)
//   k_logits_bwd_drep    dRep[b,:] = sum_n dlogit[b,n] E[n,:]   (b-chunk x item-range workgroups, slabs)
//   k_logits_bwd_de      dE[n,:]   = sum_b dlogit[b,n] rep[b,:] (item-tile workgroups; rows written once, no atomics)
// with dlogit[b,n] = w_b * (softmax_b[n] - target_b[n]) for n < ncol_b, else 0.
// Row descriptors (RowInfo): label, ncol (N for one-hot rows, Np for distilled rows), weight
// (1/B_train or lambda/B_ex, global counts under data parallelism), teacher row + teacher log-sum-exp.
#include "common.h"
#include "../../include/ader_hip.h"

#define HP 160
#define LDE 162         // (row, k) operand reads conflict-free
#define LDD 68          // 64x64 dlogit tile
#define TI 64           // items per sub-tile
#define TB 64           // batch rows per chunk
#define MAXB 1024       // max (padded) batch rows per launch

enum { MODE_LSE = 0, MODE_RANK = 1, MODE_STORE = 2 };

struct RowInfo {
    const int* lab;        // [Bp] 1-based target item, 0 = soft target only
    const int* ncol;       // [Bp] valid columns (0 for padding rows)
    const float* wrow;     // [Bp] loss weight
    const int* trow;       // [Bp] teacher row or -1
    const float* tlse;     // [Bp] teacher log-sum-exp
    const float* teacher;  // [*, ldt]
    long ldt;
};

struct LogitArgs {
    const float* rep;      // [B,H]
    const float* emb1;     // table row 1 (item 1) : [N,H]
    int B, Bp, H, N;
    RowInfo ri;
    // MODE_LSE
    float* part;           // [grid][Bp][3]
    int sub;               // 64-item sub-tiles per workgroup
    // MODE_RANK
    const float* tlogit;   // [Bp] logit of the target item (computed by the same MFMA path)
    int* rank;             // [Bp] zero-initialised
    // MODE_STORE
    float* out; long ldo;  // [B, ldo]
    // backward
    const float* lse;      // [Bp]
    float* demb1;          // gradient row of item 1 : [N,H]
    float* slab;           // drep slabs [ranges][Bp][HP]
    int ranges;
};

__device__ __forceinline__ void stage_tile(float* dst, const float* src, int row0, int nrows, int H, int tid, int nthreads) {
    // dst[r][c] = src[(row0+r)*H + c] for row0+r < nrows, c < H; zeros elsewhere.  [64][LDE]
    for (int i = tid; i < 64 * LDE; i += nthreads) {
        const int r = i / LDE, c = i - r * LDE;
        dst[i] = (row0 + r < nrows && c < H) ? src[(size_t)(row0 + r) * H + c] : 0.0f;
    }
}

__device__ __forceinline__ void merge_ml(float& m, float& l, float m2, float l2) {
    const float mn = fmaxf(m, m2);
    const float a = (m == -INFINITY) ? 0.0f : l * expf(m - mn);
    const float b = (m2 == -INFINITY) ? 0.0f : l2 * expf(m2 - mn);
    m = mn; l = a + b;
}

// C^T tile: rows = items (MFMA M), cols = batch rows (MFMA N) so that per-batch-row reductions over items
// are lane-local.  8 waves: wave (mw = w&3 -> 16 items, nw = w>>2 -> 32 batch rows).
template <int MODE>
__global__ __launch_bounds__(512) void k_logits_tile(LogitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* E_l = smem;                       // [TI][LDE]
    float* R_l = E_l + TI * LDE;             // [TB][LDE]
    float* wp = R_l + TB * LDE;              // [4][TB][3] wave partials
    float* run = wp + 4 * TB * 3;            // [MAXB][3] running (m, l, dot) per batch row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mw = wave & 3, nw = wave >> 2;
    const int H = a.H, ksteps = (H + 3) >> 2;
    const int r16 = lane & 15, q = lane >> 4;
    if (MODE == MODE_LSE)
        for (int i = tid; i < a.Bp; i += 512) { run[i * 3 + 0] = -INFINITY; run[i * 3 + 1] = 0.0f; run[i * 3 + 2] = 0.0f; }
    const int nsub = (MODE == MODE_LSE) ? a.sub : 1;
    for (int st = 0; st < nsub; ++st) {
        const int tile0 = (blockIdx.x * nsub + st) * TI;
        if (tile0 >= a.N) break;
        __syncthreads();
        stage_tile(E_l, a.emb1, tile0, a.N, H, tid, 512);
        for (int bc = 0; bc < a.Bp / TB; ++bc) {
            __syncthreads();
            stage_tile(R_l, a.rep, bc * TB, a.B, H, tid, 512);
            __syncthreads();
            f32x4 acc[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mma_tile<2>(E_l + mw * 16 * LDE, LDE, 1, R_l + nw * 32 * LDE, 1, LDE, ksteps, acc, lane);
            const int item0 = tile0 + mw * 16 + q * 4;        // this lane's 4 consecutive items
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int bl = nw * 32 + j * 16 + r16;        // batch row within the chunk
                const int b = bc * TB + bl;
                const int nc = a.ri.ncol[b];
                if (MODE == MODE_STORE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (b < a.B && item0 + r < a.N) a.out[(size_t)b * a.ldo + item0 + r] = acc[j][r];
                } else if (MODE == MODE_RANK) {
                    const int tgt = a.ri.lab[b] - 1;
                    const float tl = a.tlogit[b];
                    int cnt = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int it = item0 + r;
                        // the target never counts itself, even if k_target_logit's rounding ever differed from the tile's
                        if (it < nc && it != tgt) cnt += (acc[j][r] > tl) || (acc[j][r] == tl && it < tgt);
                    }
                    cnt += __shfl_xor(cnt, 16, 64);
                    cnt += __shfl_xor(cnt, 32, 64);
                    if (q == 0 && cnt) atomicAdd(a.rank + b, cnt);
                } else {
                    float m = -INFINITY, l = 0.0f, dot = 0.0f;
                    const int tgt = a.ri.lab[b] - 1;
                    const int tr = a.ri.trow[b];
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (item0 + r < nc) m = fmaxf(m, acc[j][r]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int it = item0 + r;
                        if (it < nc) {
                            const float s = acc[j][r];
                            l += expf(s - m);
                            if (it == tgt) dot += s;
                            if (tr >= 0) dot += expf(a.ri.teacher[(size_t)tr * a.ri.ldt + it] - a.ri.tlse[b]) * s;
                        }
                    }
#pragma unroll
                    for (int o = 16; o <= 32; o <<= 1) {
                        const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
                        merge_ml(m, l, m2, l2);
                        dot += __shfl_xor(dot, o, 64);
                    }
                    if (q == 0) { float* w = wp + (mw * TB + bl) * 3; w[0] = m; w[1] = l; w[2] = dot; }
                }
            }
            if (MODE == MODE_LSE) {
                __syncthreads();
                if (tid < TB) {
                    float* rn = run + (bc * TB + tid) * 3;
                    float m = rn[0], l = rn[1], dot = rn[2];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float* p = wp + (w * TB + tid) * 3;
                        merge_ml(m, l, p[0], p[1]);
                        dot += p[2];
                    }
                    rn[0] = m; rn[1] = l; rn[2] = dot;
                }
            }
        }
    }
    if (MODE == MODE_LSE) {
        __syncthreads();
        float* o = a.part + (size_t)blockIdx.x * a.Bp * 3;
        for (int i = tid; i < a.Bp * 3; i += 512) o[i] = run[i];
    }
}


// Row descriptors for one step: train rows first (one-hot targets, ADER.py:118-121), exemplar rows after
// (ADER.py:113-137): distilled rows (ex_trow != NULL) use ncol = Np and a teacher row; one-hot exemplar rows
// (disable_distillation, ADER.py:126-131) use their label over all N columns; padding rows get ncol = 0.
__global__ __launch_bounds__(256) void k_build_rowinfo(const int* __restrict__ pos, int n_train, const int* __restrict__ ex_pos,
                                                       const int* __restrict__ ex_trow, int n_ex, int N, int Np, float w_train, float w_ex,
                                                       int Bp, int* __restrict__ lab, int* __restrict__ ncol, float* __restrict__ wrow,
                                                       int* __restrict__ trow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp) return;
    int l = 0, nc = 0, tr = -1; float w = 0.0f;
    // A label / teacher row of 0 / -1 marks a PADDING row (data-parallel shards are padded to equal row counts): weight 0,
    // so it adds nothing to the loss or to any gradient.  Real labels are item ids >= 1 (util.py:161-171).
    if (i < n_train) { l = pos[i]; nc = N; w = (l > 0) ? w_train : 0.0f; }
    else if (i < n_train + n_ex) {
        const int e = i - n_train;
        if (ex_trow) { nc = Np; tr = ex_trow[e]; w = (tr >= 0) ? w_ex : 0.0f; if (tr < 0) tr = 0; }
        else { nc = N; l = ex_pos[e]; w = (l > 0) ? w_ex : 0.0f; }
    }
    lab[i] = l; ncol[i] = nc; wrow[i] = w; trow[i] = tr;
}

// lse[b] = logsumexp over the row's valid columns; rowloss[b] = w_b * (lse_b - sum_n target_n * logit_n).
// One wave per batch row merges the per-workgroup partials in a fixed order.
__global__ __launch_bounds__(256) void k_lse_loss(const float* __restrict__ part, int nparts, int Bp, int B,
                                                  const float* __restrict__ wrow, float* __restrict__ lse, float* __restrict__ rowloss) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= Bp) return;
    float m = -INFINITY, l = 0.0f, dot = 0.0f;
    for (int p = lane; p < nparts; p += 64) {
        const float* x = part + ((size_t)p * Bp + b) * 3;
        merge_ml(m, l, x[0], x[1]);
        dot += x[2];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
        merge_ml(m, l, m2, l2);
        dot += __shfl_xor(dot, o, 64);
    }
    if (lane == 0) {
        const float z = (b < B && l > 0.0f) ? m + logf(l) : 0.0f;
        lse[b] = z;
        rowloss[b] = (b < B) ? wrow[b] * (z - dot) : 0.0f;
    }
}

// out[0] = sum_i x[i] (single workgroup, fixed tree: deterministic)
__global__ __launch_bounds__(256) void k_sum(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) acc += x[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// Row-wise log-sum-exp of the teacher logits (softmax(exemplar_logits), ADER.py:135): one workgroup per row.
__global__ __launch_bounds__(256) void k_row_lse(const float* __restrict__ x, long ld, int ncols, const int* __restrict__ rows,
                                                 float* __restrict__ out) {
    __shared__ float red[256];
    const int r = rows ? rows[blockIdx.x] : blockIdx.x;
    if (r < 0) { if (threadIdx.x == 0) out[blockIdx.x] = 0.0f; return; }
    const float* p = x + (size_t)r * ld;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < ncols; i += 256) m = fmaxf(m, p[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
    m = red[0];
    __syncthreads();
    float l = 0.0f;
    for (int i = threadIdx.x; i < ncols; i += 256) l += expf(p[i] - m);
    red[threadIdx.x] = l;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) out[blockIdx.x] = m + logf(red[0]);
}

__device__ __forceinline__ float dlogit(const LogitArgs& a, int b, int it, float s, int nc, float w, float z, int tgt, int tr, float tz) {
    if (it >= nc) return 0.0f;
    float p = expf(s - z);
    if (it == tgt) p -= 1.0f;
    if (tr >= 0) p -= expf(a.ri.teacher[(size_t)tr * a.ri.ldt + it] - tz);
    return w * p;
}

// dE tile: one workgroup per 64 items, loops over all batch chunks; wave (mw, nw) owns dE rows 16mw.. x columns 80nw..
__global__ __launch_bounds__(512) void k_logits_bwd_de(LogitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* E_l = smem;
    float* R_l = E_l + TI * LDE;
    float* D_l = R_l + TB * LDE;             // [items][batch] dlogit^T
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mw = wave & 3, nw = wave >> 2;
    const int H = a.H, ksteps = (H + 3) >> 2;
    const int r16 = lane & 15, q = lane >> 4;
    const int tile0 = blockIdx.x * TI;
    stage_tile(E_l, a.emb1, tile0, a.N, H, tid, 512);
    f32x4 dacc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) dacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int bc = 0; bc < a.Bp / TB; ++bc) {
        __syncthreads();
        stage_tile(R_l, a.rep, bc * TB, a.B, H, tid, 512);
        __syncthreads();
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<2>(E_l + mw * 16 * LDE, LDE, 1, R_l + nw * 32 * LDE, 1, LDE, ksteps, acc, lane);
        const int il = mw * 16 + q * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int bl = nw * 32 + j * 16 + r16;
            const int b = bc * TB + bl;
            const int nc = a.ri.ncol[b], tgt = a.ri.lab[b] - 1, tr = a.ri.trow[b];
            const float w = a.ri.wrow[b], z = a.lse[b], tz = a.ri.tlse[b];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                D_l[(il + r) * LDD + bl] = dlogit(a, b, tile0 + il + r, acc[j][r], nc, w, z, tgt, tr, tz);
        }
        __syncthreads();
        mma_tile<5>(D_l + mw * 16 * LDD, LDD, 1, R_l + nw * 80, LDE, 1, TB / 4, dacc, lane);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int h = nw * 80 + j * 16 + r16;
        if (h >= H) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int it = tile0 + mw * 16 + q * 4 + r;
            if (it < a.N) a.demb1[(size_t)it * H + h] = dacc[j][r];
        }
    }
}

// dRep: workgroup = (batch chunk, item range); dRep chunk accumulated in registers over the range, written as a slab.
// Block -> (range, chunk) mapping keeps the chunks that stream the same item range on one XCD (blocks b and b+8
// share an XCD) so the table rows are re-read from that XCD's L2 rather than HBM.  Speed only, not correctness.
__global__ __launch_bounds__(512) void k_logits_bwd_drep(LogitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* E_l = smem;
    float* R_l = E_l + TI * LDE;
    float* D_l = R_l + TB * LDE;             // [batch][items] dlogit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mw = wave & 3, nw = wave >> 2;
    const int H = a.H, ksteps = (H + 3) >> 2;
    const int r16 = lane & 15, q = lane >> 4;
    const int nchunk = a.Bp / TB;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range = xcd + 8 * (slot / nchunk), bc = slot % nchunk;
    if (range >= a.ranges) return;
    const int nsub_total = (a.N + TI - 1) / TI;
    const int per = (nsub_total + a.ranges - 1) / a.ranges;
    const int s_begin = range * per, s_end = min(nsub_total, s_begin + per);
    stage_tile(R_l, a.rep, bc * TB, a.B, H, tid, 512);
    f32x4 racc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) racc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // row constants of this lane's 4 batch rows (D rows = (lane>>4)*4 + r)
    int nc[4], tgt[4], tr[4]; float w[4], z[4], tz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = bc * TB + mw * 16 + q * 4 + r;
        nc[r] = a.ri.ncol[b]; tgt[r] = a.ri.lab[b] - 1; tr[r] = a.ri.trow[b];
        w[r] = a.ri.wrow[b]; z[r] = a.lse[b]; tz[r] = a.ri.tlse[b];
    }
    for (int s = s_begin; s < s_end; ++s) {
        const int tile0 = s * TI;
        __syncthreads();
        stage_tile(E_l, a.emb1, tile0, a.N, H, tid, 512);
        __syncthreads();
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_tile<2>(R_l + mw * 16 * LDE, LDE, 1, E_l + nw * 32 * LDE, 1, LDE, ksteps, acc, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int il = nw * 32 + j * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = bc * TB + mw * 16 + q * 4 + r;
                D_l[(mw * 16 + q * 4 + r) * LDD + il] = dlogit(a, b, tile0 + il, acc[j][r], nc[r], w[r], z[r], tgt[r], tr[r], tz[r]);
            }
        }
        __syncthreads();
        mma_tile<5>(D_l + mw * 16 * LDD, LDD, 1, E_l + nw * 80, LDE, 1, TI / 4, racc, lane);
    }
    float* o = a.slab + ((size_t)range * a.Bp + bc * TB) * HP;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o[(size_t)(mw * 16 + q * 4 + r) * HP + nw * 80 + j * 16 + r16] = racc[j][r];
}

// Logit of each row's target item through the same MFMA k-order as the tile kernels (so the == comparison of the
// rank kernel is exact): 64 batch rows x their 64 gathered target rows, diagonal extracted.
__global__ __launch_bounds__(256) void k_target_logit(LogitArgs a, float* __restrict__ tl) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* E_l = smem;
    float* R_l = E_l + TI * LDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, ksteps = (H + 3) >> 2;
    const int bc = blockIdx.x;
    for (int i = tid; i < 64 * LDE; i += 256) {
        const int r = i / LDE, c = i - r * LDE;
        const int b = bc * TB + r;
        const int t = (b < a.B) ? a.ri.lab[b] - 1 : -1;
        E_l[i] = (t >= 0 && t < a.N && c < H) ? a.emb1[(size_t)t * H + c] : 0.0f;
        R_l[i] = (b < a.B && c < H) ? a.rep[(size_t)b * H + c] : 0.0f;
    }
    __syncthreads();
    // wave w: items (targets) 16w..16w+15 x batch rows 16w..16w+15 (only the diagonal block is needed)
    f32x4 acc[1];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma_tile<1>(E_l + wave * 16 * LDE, LDE, 1, R_l + wave * 16 * LDE, 1, LDE, ksteps, acc, lane);
    const int col = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (q * 4 + r == col) { const int b = bc * TB + wave * 16 + col; tl[b] = acc[0][r]; }
}

// ============================================================================================= C ABI
static const size_t kTileLds = (size_t)(2 * 64 * LDE + 4 * TB * 3 + MAXB * 3) * sizeof(float);
static const size_t kBwdLds = (size_t)(2 * 64 * LDE + 64 * LDD) * sizeof(float);
static const size_t kTgtLds = (size_t)(2 * 64 * LDE) * sizeof(float);

template <typename K>
static int set_lds(K kern, size_t bytes, bool& flag) {
    if (!flag) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        flag = true;
    }
    return 0;
}

static int fill_args(LogitArgs& a, const float* rep, const float* emb, int B, int Bp, int H, int N, const int* lab, const int* ncol,
                     const float* wrow, const int* trow, const float* tlse, const float* teacher, long ldt) {
    if (Bp % TB != 0 || Bp > MAXB || B > Bp || H > HP || H < 1) return -2;
    a.rep = rep; a.emb1 = emb + H; a.B = B; a.Bp = Bp; a.H = H; a.N = N;
    a.ri.lab = lab; a.ri.ncol = ncol; a.ri.wrow = wrow; a.ri.trow = trow; a.ri.tlse = tlse; a.ri.teacher = teacher; a.ri.ldt = ldt;
    a.part = nullptr; a.sub = 1; a.tlogit = nullptr; a.rank = nullptr; a.out = nullptr; a.ldo = 0; a.lse = nullptr;
    a.demb1 = nullptr; a.slab = nullptr; a.ranges = 0;
    return 0;
}

extern "C" {

// Sub-tiles per workgroup / number of partial slabs for a catalog of N items.
int ader_logits_sub(int N) {
    const int nsub = (N + TI - 1) / TI;
    int sub = nsub / 2048;
    if (sub < 1) sub = 1;
    if (sub > 8) sub = 8;
    return sub;
}
int ader_logits_parts(int N) {
    const int nsub = (N + TI - 1) / TI;
    const int sub = ader_logits_sub(N);
    return (nsub + sub - 1) / sub;
}
int ader_logits_ranges(int N, int Bp) {
    const int nsub = (N + TI - 1) / TI;
    int target = 1024 / (Bp / TB);          // ~1024 workgroups
    int r = 8;
    while (r * 2 <= target && r * 2 <= nsub) r *= 2;
    return r;                                // multiple of 8
}

int ader_build_rowinfo(const int* pos, int n_train, const int* ex_pos, const int* ex_trow, int n_ex, int N, int Np, float w_train,
                       float w_ex, int Bp, int* lab, int* ncol, float* wrow, int* trow, void* stream) {
    if (Bp <= 0) return 0;
    hipLaunchKernelGGL(k_build_rowinfo, dim3((Bp + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, n_train, ex_pos, ex_trow, n_ex, N,
                       Np, w_train, w_ex, Bp, lab, ncol, wrow, trow);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_row_lse(const float* x, long ld, int ncols, const int* rows, int nrows, float* out, void* stream) {
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(k_row_lse, dim3(nrows), dim3(256), 0, (hipStream_t)stream, x, ld, ncols, rows, out);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Forward: lse[Bp], rowloss[Bp], loss[1].  part: ader_logits_parts(N) * Bp * 3 floats of scratch.
int ader_logits_loss_fwd(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* lab, const int* ncol,
                         const float* wrow, const int* trow, const float* tlse, const float* teacher, long ldt, float* part,
                         float* lse, float* rowloss, float* loss, void* stream) {
    if (B <= 0) return 0;
    LogitArgs a;
    int rc = fill_args(a, rep, emb, B, Bp, H, N, lab, ncol, wrow, trow, tlse, teacher, ldt);
    if (rc) return rc;
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
    rc = set_lds(k_logits_tile<MODE_LSE>, kTileLds, f);
    if (rc) return rc;
    a.part = part; a.sub = ader_logits_sub(N);
    const int parts = ader_logits_parts(N);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_logits_tile<MODE_LSE>, dim3(parts), dim3(512), kTileLds, st, a);
    hipLaunchKernelGGL(k_lse_loss, dim3((Bp + 3) / 4), dim3(256), 0, st, part, parts, Bp, B, wrow, lse, rowloss);
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, rowloss, B, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Backward, part 1: dRep [B,H] (slab: ader_logits_ranges(N,Bp) * Bp * 160 floats of scratch).
int ader_logits_bwd_drep(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* lab, const int* ncol,
                         const float* wrow, const int* trow, const float* tlse, const float* teacher, long ldt, const float* lse,
                         float* slab, float* drep, void* stream) {
    if (B <= 0) return 0;
    LogitArgs a;
    int rc = fill_args(a, rep, emb, B, Bp, H, N, lab, ncol, wrow, trow, tlse, teacher, ldt);
    if (rc) return rc;
    static bool f1_dev[ADER_MAX_DEV] = {};
    bool& f1 = f1_dev[ader_cur_dev()];
    rc = set_lds(k_logits_bwd_drep, kBwdLds, f1);
    if (rc) return rc;
    a.lse = lse; a.slab = slab; a.ranges = ader_logits_ranges(N, Bp);
    const int nchunk = Bp / TB;
    hipLaunchKernelGGL(k_logits_bwd_drep, dim3(a.ranges * nchunk), dim3(512), kBwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return ader_reduce_slabs(slab, (long)Bp * HP, a.ranges, HP, B, H, drep, nullptr, stream);
}

// Backward, part 2: table gradient rows 1..N (overwritten; every row written exactly once, no atomics).
int ader_logits_bwd_demb(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* lab, const int* ncol,
                         const float* wrow, const int* trow, const float* tlse, const float* teacher, long ldt, const float* lse,
                         float* demb, void* stream) {
    if (B <= 0) return 0;
    LogitArgs a;
    int rc = fill_args(a, rep, emb, B, Bp, H, N, lab, ncol, wrow, trow, tlse, teacher, ldt);
    if (rc) return rc;
    static bool f2_dev[ADER_MAX_DEV] = {};
    bool& f2 = f2_dev[ader_cur_dev()];
    rc = set_lds(k_logits_bwd_de, kBwdLds, f2);
    if (rc) return rc;
    a.lse = lse; a.demb1 = demb + H;
    hipLaunchKernelGGL(k_logits_bwd_de, dim3((N + TI - 1) / TI), dim3(512), kBwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Dense logits out[b, 0:N] = rep[b] . E[1..N]^T   (ADER.py:92; exemplar teacher logits, util.py:433)
int ader_logits_store(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* ncol_all, float* out, long ldo,
                      void* stream) {
    if (B <= 0) return 0;
    LogitArgs a;
    int rc = fill_args(a, rep, emb, B, Bp, H, N, nullptr, ncol_all, nullptr, nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
    rc = set_lds(k_logits_tile<MODE_STORE>, kTileLds, f);
    if (rc) return rc;
    a.out = out; a.ldo = ldo;
    hipLaunchKernelGGL(k_logits_tile<MODE_STORE>, dim3((N + TI - 1) / TI), dim3(512), kTileLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

// rank[b] = #{n < N : logit[b,n] > logit[b,t]  or (== and n < t)},  t = target[b]-1   (0-based rank, ties -> lower index first)
// tlogit: Bp floats scratch; ncol: [Bp] = N for real rows, 0 for padding; rank: [Bp] (zeroed here).
int ader_rank_targets(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* target, const int* ncol,
                      float* tlogit, int* rank, void* stream) {
    if (B <= 0) return 0;
    LogitArgs a;
    int rc = fill_args(a, rep, emb, B, Bp, H, N, target, ncol, nullptr, nullptr, nullptr, nullptr, 0);
    if (rc) return rc;
    static bool f1_dev[ADER_MAX_DEV] = {}, f2_dev[ADER_MAX_DEV] = {};
    bool& f1 = f1_dev[ader_cur_dev()];
    bool& f2 = f2_dev[ader_cur_dev()];
    rc = set_lds(k_logits_tile<MODE_RANK>, kTileLds, f1);
    if (rc) return rc;
    rc = set_lds(k_target_logit, kTgtLds, f2);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(rank, 0, sizeof(int) * (size_t)Bp, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_target_logit, dim3(Bp / TB), dim3(256), kTgtLds, st, a, tlogit);
    a.tlogit = tlogit; a.rank = rank;
    hipLaunchKernelGGL(k_logits_tile<MODE_RANK>, dim3((N + TI - 1) / TI), dim3(512), kTileLds, st, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
