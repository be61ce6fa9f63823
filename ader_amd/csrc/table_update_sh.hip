// Fused table update, bf16 mode: table-gradient GEMM + sparse terms + TF-Adam (ADER.py:96) + bf16 shadow rows in one pass.
// The GEMM operand E is the tile's bf16 SHADOW rows (336 B per row, read once), theta / m / v are streamed in rounds of 16-byte
// vectors.  64-row tiles on v_mfma_f32_16x16x32_bf16, three workgroups per CU (k_tab16 below).  This form reads 8 % more bytes
// than table_update.hip's theta-resident kernel (the shadow rows) but is 10 % faster: both are bound by the per-workgroup
// latency chain and the HBM queue, and resident waves per CU are what buys time (round-2 measurements in DESIGN.md section 6;
// the 128-row 32x32x16 predecessor of this kernel fitted two workgroups per CU: 0.95 ms against 0.90 ms per 10^6 rows).
// table_update.hip serves the x3 mode and the gradient-only entry point.
#include "lbf_common.h"
#include "../../include/ader_hip.h"

struct ShArgs {
    int tile_off;               // first 128-item tile handled by this launch (row-sharded table update)
    const bf16* sh1;            // bf16 shadow of the table, row of item 1: rows of LDR elements (336 B), cols >= H zero
    int vrows;                  // shadow rows available from sh1 (= item_num)
    const bf16* rep_bf;         // [Bp][LDR]
    int B, Bp, H, N, ranges;
    float* pm; float* pl; float* pO;    // [ranges][Bp], [ranges][Bp], [ranges][Bp][HP]
    const float* off;           // [Bp] log2(w_b) - lse2_b
    float* demb1;               // gradient row of item 1
    // KD rows (ADER.py:132-137): batch rows [kd_row0, Bp) are distilled exemplar rows: dlogit = w (softmax(s[:Np]) - softmax(t)),
    // zero for items >= Np.  kd_row0 % 128 == 0; = Bp: none
    int kd_row0, Np;
    const float* teacher; long ldt; const int* trow; const float* tlse2;     // as in ader_lbf_fwd_kd ([Bp] arrays)
};
#define PCS_ROW (LDR * 2 / 16)     // 16-byte pieces per shadow row (21)

// sparse terms and optimiser constants (the lists are addressed through their 64-id bucket offsets)
struct FuseArgs128 {
    const int* sp_ids; const int* sp_rows; int n_sp; const float* sp_src; float sp_scale;   // input-embedding rows (sorted by id)
    const int* tg_ids; const int* tg_rows; int n_tg; const float* wrow;                      // one-hot targets (sorted by id)
    const int* sp_start; const int* tg_start;   // bucket offsets into the two lists: bucket j = ids [64j + 1, 64j + 65)
    float* emb1; float* m1; float* v1; bf16* sh1w;                                           // row of item 1 of theta/m/v/shadow
    float lr_t, omb1, omb2, eps;
    const float* extra1;        // EXTRA: dense gradient rows to add (row of item 1; [.,H] fp32), e.g. distilled rows' term
};

#define SPV 3                      // input-embedding gradient rows prefetched under the GEMM phase
#define HEAVY_N 32                 // a bucket with more entries than this in either list takes the heavy path
#define HVB 16                     // gradient rows in flight per thread on the heavy path
#define SPB 8                      // sparse-list entries per batch of the optimiser phase (loads of a batch are independent)
#define AV 6                       // 16-byte vectors per thread and load round of the optimiser phase (x theta, m, v)

// rep chunk (64 rows x 336 B, contiguous) global -> registers -> LDS buffer, one chunk ahead of the MFMAs
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
// one round of theta / m / v vectors of the tile's flat [64*H] block: all loads issued before any math or store; (row, col) of a
// vector -- needed only for the bf16 shadow row -- is stepped without divisions
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

// ---------------------------------------------------------------------------------------------------------------------------
// 64-row tiles on v_mfma_f32_16x16x32_bf16: each of the 4 waves owns 16 table rows, so the dE accumulators are 40 registers per
// lane instead of 80 and the kernel fits THREE workgroups per CU (<= 168 registers, 45 KB of LDS) instead of two.  The update is
// bound by its per-workgroup latency chain (tile load -> GEMM over all batch rows -> theta/m/v rounds), so resident waves per CU
// are what buys time: one workgroup per CU takes 1.37 ms per 10^6 rows, two take 0.87 ms (measured with an LDS pad).
//   S block  = 16 batch rows x 16 items: A = rep rows (lane: row c16, k = 8g..8g+7), B = this wave's E fragments (5 k-steps of 32)
//   P^T.rep  = 16 items x 16 channels, K = 32 batch rows: the A fragment of lane (c16, g) is its own p values of TWO S blocks
//              (rows 4g..4g+3 of block A, then of block B -- no lane movement); the B fragment reads exactly those rows k-major
//              with two ds_read_b64_tr_b16 per MFMA.
// No cross-wave reduction: a wave accumulates the whole batch for its 16 rows.  The optimiser phase walks the tile's rows of
// theta / m / v as one flat block of 16-byte vectors.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16_bf16(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <bool EXTRA, bool KD>
__global__ __launch_bounds__(256, 3) void k_tab16(ShArgs a, FuseArgs128 f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R_l = (bf16*)smem_raw;                        // [2][64][LDR]  (first the table tile, last the dE staging tile)
    float* off_l = (float*)(smem_raw + 2 * 64 * LDR * sizeof(bf16));   // [Bp]
    int* meta_l = (int*)(off_l + a.Bp);                 // per list: [k0, k1, 8 x (id, row)] = 18 ints, 2 lists
    float* toff_l = (float*)(meta_l + 2 * 18);          // KD: [Bp - kd_row0]
    int* trow_l = (int*)(toff_l + (a.Bp - a.kd_row0));  // KD: [Bp - kd_row0]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int H = a.H, N = a.N;
    const int tile0 = (blockIdx.x + a.tile_off) * 64;
    const int it0 = tile0 + wave * 16;
    {   // table tile: 64 shadow rows, contiguous -> LDS (coalesced 16-B pieces) -> operand fragments in registers
        const uint4* src = (const uint4*)(a.sh1 + (size_t)tile0 * LDR);
        uint4* dst = (uint4*)R_l;
        for (int idx = tid; idx < 64 * PCS_ROW; idx += 256) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (tile0 + idx / PCS_ROW < a.vrows) v = src[idx];
            dst[idx] = v;
        }
    }
    for (int i = tid; i < a.Bp; i += 256) off_l[i] = a.off[i];
    if (KD) {
        for (int i = tid; i < a.Bp - a.kd_row0; i += 256) {
            const int b = a.kd_row0 + i, tr = a.trow[b];
            const float w = f.wrow[b];
            toff_l[i] = (tr >= 0 && w > 0.0f) ? log2f(w) - a.tlse2[b] : -INFINITY;
            trow_l[i] = tr < 0 ? 0 : tr;
        }
    }
    if (tid < 2) {
        // the tile's two sparse lists (bucket bounds and first entries), fetched under the GEMM phase
        const int bkt = tile0 >> 6;
        const int* st = tid ? f.tg_start : f.sp_start;
        const int* ids = tid ? f.tg_ids : f.sp_ids;
        const int* rows = tid ? f.tg_rows : f.sp_rows;
        int* mt = meta_l + tid * 18;
        int k0 = 0, k1 = 0;
        if (tile0 < N) { k0 = st[bkt]; k1 = st[bkt + 1]; }
        mt[0] = k0; mt[1] = k1;
        for (int i = 0; i < 8 && k0 + i < k1; ++i) { mt[2 + 2 * i] = ids[k0 + i]; mt[3 + 2 * i] = rows[k0 + i]; }
    }
    __syncthreads();
    float spv[SPV];
#pragma unroll
    for (int i = 0; i < SPV; ++i)
        spv[i] = (tid < H && meta_l[0] + i < meta_l[1]) ? f.sp_src[(size_t)meta_l[3 + 2 * i] * H + tid] * f.sp_scale : 0.0f;
    bf16x8 efrag[5];                                    // lane (item c16, k-group g) holds E[item][32 ks + 8 g + 0..7]
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) efrag[ks] = *(const bf16x8*)(R_l + (wave * 16 + c16) * LDR + 32 * ks + 8 * g);
    __syncthreads();
    f32x4v dE[10];
#pragma unroll
    for (int cb = 0; cb < 10; ++cb) dE[cb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const int nch = a.Bp >> 6;
    const int n16 = 64 * LDR * 2 / 16;                  // 16-byte pieces per 64-row chunk (1344)
    uint4 pf[6];
    LBF_RPREFETCH(0); LBF_RSTAGE(0);
    __syncthreads();
    int cur = 0;
    const int q4 = c16 >> 2, p4 = c16 & 3;
    for (int c = 0; c < nch; ++c) {
        const bool more = c + 1 < nch;
        if (more) LBF_RPREFETCH(c + 1);
        const bf16* Rb = R_l + cur * 64 * LDR;
        const int b0 = c * 64;
        // KD rows: this lane's 16 teacher logits (item it0 + c16, batch rows b0 + 16 rb + 4 g + reg), requested ahead of the MFMAs
        float tv[KD ? 16 : 1];
        const bool kdc = KD && b0 >= a.kd_row0;         // (workgroup-uniform: chunks do not straddle kd_row0)
        if (kdc && it0 + c16 < a.Np) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                tv[KD ? j : 0] = a.teacher[(size_t)trow_l[b0 - a.kd_row0 + 16 * (j >> 2) + 4 * g + (j & 3)] * a.ldt + it0 + c16];
        }
        f32x4v S[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) S[rb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const bf16x8 af = *(const bf16x8*)(Rb + (rb * 16 + c16) * LDR + 32 * ks + 8 * g);
                S[rb] = mfma16_bf16(af, efrag[ks], S[rb]);
            }
        }
        // rows of S are batch rows: p = w_b * softmax = exp2(S*log2e + off_b)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const float4 o4 = *(const float4*)(off_l + b0 + 16 * rb + 4 * g);
            S[rb][0] = __builtin_amdgcn_exp2f(fmaf(S[rb][0], LOG2E, o4.x));
            S[rb][1] = __builtin_amdgcn_exp2f(fmaf(S[rb][1], LOG2E, o4.y));
            S[rb][2] = __builtin_amdgcn_exp2f(fmaf(S[rb][2], LOG2E, o4.z));
            S[rb][3] = __builtin_amdgcn_exp2f(fmaf(S[rb][3], LOG2E, o4.w));
        }
        if (kdc) {              // dlogit of a distilled row: w (softmax(s[:Np]) - softmax(t)) for items < Np, 0 beyond
            if (it0 + c16 < a.Np) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    S[j >> 2][j & 3] -= __builtin_amdgcn_exp2f(fmaf(tv[KD ? j : 0], LOG2E,
                                                                    toff_l[b0 - a.kd_row0 + 16 * (j >> 2) + 4 * g + (j & 3)]));
            } else {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) S[rb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            }
        }
        bf16x8 pA, pB;          // k order of a fragment: rows 4g..4g+3 of the first S block, then of the second
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pA[j] = (bf16)S[0][j]; pA[4 + j] = (bf16)S[1][j];
            pB[j] = (bf16)S[2][j]; pB[4 + j] = (bf16)S[3][j];
        }
#pragma unroll
        for (int cb = 0; cb < 10; ++cb) {
            const bf16* base = Rb + (4 * g + q4) * LDR + 16 * cb + 4 * p4;
            const bf16x4 t0 = tr_read(base), t1 = tr_read(base + 16 * LDR);
            const bf16x4 t2 = tr_read(base + 32 * LDR), t3 = tr_read(base + 48 * LDR);
            bf16x8 bA, bB;
#pragma unroll
            for (int j = 0; j < 4; ++j) { bA[j] = t0[j]; bA[4 + j] = t1[j]; bB[j] = t2[j]; bB[4 + j] = t3[j]; }
            dE[cb] = mfma16_bf16(pA, bA, dE[cb]);
            dE[cb] = mfma16_bf16(pB, bB, dE[cb]);
        }
        if (more) LBF_RSTAGE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // ---- optimiser phase: the tile's rows are ONE contiguous block of 64*H floats in theta / m / v (and in F_l)
    float* F_l = (float*)smem_raw;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const int rows_valid = min(64, N - tile0);
    const int n_el = rows_valid > 0 ? rows_valid * H : 0;
    float* __restrict__ gp = f.emb1 + (size_t)tile0 * H;
    float* __restrict__ gm = f.m1 + (size_t)tile0 * H;
    float* __restrict__ gv = f.v1 + (size_t)tile0 * H;
    const int head = (((uintptr_t)gp) & 15) ? 2 : 0;
    int e = head + 4 * tid;
    int row = e / H, col = e - row * H;
    const int step_r = 1024 / H, step_c = 1024 - step_r * H;
    f32x4_t P[AV], M[AV], V[AV], G[EXTRA ? AV : 1];
    const float* __restrict__ gx = EXTRA ? f.extra1 + (size_t)tile0 * H : nullptr;
    int E[AV], RC[AV], NV[AV];
    // A bucket that holds a hot item (Zipf ids: hundreds of entries) takes the HEAVY path below: its (id, row) lists are fetched
    // cooperatively, 256 entries per round trip, and the gradient rows HVB at a time -- with the optimiser loads requested AFTER
    // the sparse terms, so that the registers are free for the deeper batches (a workgroup-uniform choice; rare tiles).
    const bool heavy = (meta_l[1] - meta_l[0] > HEAVY_N) || (meta_l[19] - meta_l[18] > HEAVY_N);
    if (!heavy) { ROUND_LOAD(); }   // first round of theta/m/v: requested BEFORE the dE staging and the sparse terms
    lds_only_barrier();
#pragma unroll
    for (int cb = 0; cb < 10; ++cb) {
        const int h = 16 * cb + c16;
        if (h < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) F_l[(wave * 16 + 4 * g + j) * H + h] = dE[cb][j];
        }
    }
    lds_only_barrier();
    {
        // sparse terms of the tile: item ids [tile0+1, tile0+65).  Thread c owns column c of every row.
        const int id_lo = tile0 + 1, id_hi = min(tile0 + 64, N) + 1;
        if (heavy) {
            int* hv_l = (int*)(smem_raw + 64 * HP * sizeof(float));     // [2][256] (id, row) of the current chunk, behind F_l
            // entries in list order (the same order, hence the same rounding, as the light path)
#define HEAVY_LIST(K0_, K1_, IDS_, ROWS_, VAL_, OP_)                                                       \
            for (int base_ = (K0_); base_ < (K1_); base_ += 256) {                                         \
                const int n_ = min(256, (K1_) - base_);                                                    \
                if (tid < n_) { hv_l[tid] = (IDS_)[base_ + tid]; hv_l[256 + tid] = (ROWS_)[base_ + tid]; } \
                __syncthreads();                                                                           \
                if (tid < H && id_lo < id_hi) {                                                            \
                    for (int e0_ = 0; e0_ < n_; e0_ += HVB) {                                              \
                        int idv[HVB];                                                                      \
                        float val[HVB];                                                                    \
                        _Pragma("unroll") for (int u = 0; u < HVB; ++u) {                                  \
                            const bool in_ = e0_ + u < n_;                                                 \
                            idv[u] = in_ ? hv_l[e0_ + u] : 0x7fffffff;                                     \
                            const int rw = in_ ? hv_l[256 + e0_ + u] : 0;                                  \
                            val[u] = (VAL_) * ((idv[u] < id_hi) ? 1.0f : 0.0f);                            \
                        }                                                                                  \
                        _Pragma("unroll") for (int u = 0; u < HVB; ++u)                                    \
                            if (idv[u] < id_hi) F_l[(idv[u] - id_lo) * H + tid] OP_ val[u];                \
                    }                                                                                      \
                }                                                                                          \
                __syncthreads();                                                                           \
            }
            HEAVY_LIST(meta_l[0], meta_l[1], f.sp_ids, f.sp_rows, f.sp_src[(size_t)rw * H + tid] * f.sp_scale, +=)
            HEAVY_LIST(meta_l[18], meta_l[19], f.tg_ids, f.tg_rows, f.wrow[rw] * (float)a.rep_bf[(size_t)rw * LDR + tid], -=)
#undef HEAVY_LIST
        } else if (tid < H && id_lo < id_hi) {
            const int* ms = meta_l;
            const int* mg = meta_l + 18;
            const int k0s = ms[0], k1s = ms[1];
#pragma unroll
            for (int i = 0; i < SPV; ++i) {                  // rows already in registers (same (id, row) order)
                if (k0s + i < k1s) {
                    const int id = ms[2 + 2 * i];
                    if (id < id_hi) F_l[(id - id_lo) * H + tid] += spv[i];
                }
            }
            // entries SPV..7 of the bucket are in the LDS record, the rest in the global lists.  Batches of SPB entries: ids and
            // rows first, then every gradient row, then the adds in entry order (the order fixes the rounding) -- a hot item's
            // bucket holds hundreds of entries, and one dependent memory round trip per ENTRY made its workgroup the straggler of
            // the launch (Zipf ids: 1.02 ms against 0.84 ms for uniform ids).  The loads are UNCONDITIONAL (row 0 for entries that do
            // not count): under a per-entry branch hipcc waits for each load at the end of its branch -- one round trip per entry
            for (int k = k0s + SPV, i = SPV; k < k1s; k += SPB, i += SPB) {
                int idv[SPB], rw[SPB];
                float val[SPB];
#pragma unroll
                for (int u = 0; u < SPB; ++u) {
                    const int ic = (i + u) < 8 ? (i + u) : 7;
                    const int id_c = ms[2 + 2 * ic], row_c = ms[3 + 2 * ic];     // the first 8 entries: from the LDS record
                    const bool in = k + u < k1s;
                    int id_g = 0, row_g = 0;
                    if (i + SPB > 8) {                           // (batch-uniform) later entries: from the global lists,
                        const int ke = in ? k + u : k0s;         //  UNCONDITIONAL loads of an always-valid entry
                        id_g = f.sp_ids[ke]; row_g = f.sp_rows[ke];
                    }
                    idv[u] = !in ? 0x7fffffff : ((i + u < 8) ? id_c : id_g);
                    rw[u] = !in ? 0 : ((i + u < 8) ? row_c : row_g);
                }
#pragma unroll
                for (int u = 0; u < SPB; ++u)      // (ids beyond max_item have no table row)
                    val[u] = f.sp_src[(size_t)rw[u] * H + tid] * ((idv[u] < id_hi) ? f.sp_scale : 0.0f);
#pragma unroll
                for (int u = 0; u < SPB; ++u)
                    if (idv[u] < id_hi) F_l[(idv[u] - id_lo) * H + tid] += val[u];
            }
            for (int k = mg[0], k1 = mg[1], i = 0; k < k1; k += SPB, i += SPB) {
                int idv[SPB], bw[SPB];
                float val[SPB];
#pragma unroll
                for (int u = 0; u < SPB; ++u) {
                    const int ic = (i + u) < 8 ? (i + u) : 7;
                    const int id_c = mg[2 + 2 * ic], b_c = mg[3 + 2 * ic];
                    const bool in = k + u < k1;
                    int id_g = 0, b_g = 0;
                    if (i + SPB > 8) {
                        const int ke = in ? k + u : mg[0];
                        id_g = f.tg_ids[ke]; b_g = f.tg_rows[ke];
                    }
                    idv[u] = !in ? 0x7fffffff : ((i + u < 8) ? id_c : id_g);
                    bw[u] = !in ? 0 : ((i + u < 8) ? b_c : b_g);
                }
#pragma unroll
                for (int u = 0; u < SPB; ++u)
                    val[u] = f.wrow[bw[u]] * (float)a.rep_bf[(size_t)bw[u] * LDR + tid] * ((idv[u] < id_hi) ? 1.0f : 0.0f);
#pragma unroll
                for (int u = 0; u < SPB; ++u)
                    if (idv[u] < id_hi) F_l[(idv[u] - id_lo) * H + tid] -= val[u];
            }
        }
    }
    if (heavy) { ROUND_LOAD(); }
    lds_only_barrier();
    bf16* __restrict__ psh = f.sh1w + (size_t)tile0 * LDR;
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

static size_t bwd_lds(int Bp, int Bk) { return (size_t)2 * 64 * LDR * sizeof(bf16) + (size_t)Bp * sizeof(float) + 2 * 18 * sizeof(int) + (size_t)Bk * 8; }

extern "C" {

// Fused bf16-mode table update over 128-row tiles (see ader_tab_update for the arguments; here the sorted lists are addressed
// through their 64-id bucket offsets sp_start / tg_start directly, and `shadow` is both the GEMM operand and rewritten).
static int tab_update_sh(const void* rep_bf, void* shadow, int item_num, int B, int Bp, int H, int N, const float* off,
                         const int* sp_ids, const int* sp_rows, const int* sp_start, int n_sp, const float* sp_src, float sp_scale,
                         const int* tg_ids, const int* tg_rows, const int* tg_start, int n_tg, const float* wrow, float* emb,
                         float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                         int tile_count, const float* extra_grad, int kd_row0, int Np, const float* teacher, long ldt, const int* trow,
                         const float* tlse2, void* stream) {
    if (B <= 0) return 0;
    if (Bp % 128 != 0 || B > Bp || H > HP || (H & 1) || H < 2 || N > item_num || !shadow) return -2;
    const bool kd = kd_row0 < Bp;
    if (kd && (kd_row0 % 128 != 0 || extra_grad || !teacher || !trow || !tlse2 || Np < 1 || Np > N)) return -2;
    static int lds_set_dev[ADER_MAX_DEV] = {};
    int& lds_set = lds_set_dev[ader_cur_dev()];
    const size_t lds = bwd_lds(Bp, kd ? Bp - kd_row0 : 0);
    if ((int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_tab16<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)k_tab16<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)k_tab16<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = (int)lds;
    }
    ShArgs a;
    a.sh1 = (const bf16*)shadow + LDR; a.vrows = item_num; a.tile_off = 0;
    a.rep_bf = (const bf16*)rep_bf; a.B = B; a.Bp = Bp; a.H = H; a.N = N; a.ranges = 0;
    a.pm = a.pl = a.pO = nullptr; a.off = off; a.demb1 = nullptr;
    a.kd_row0 = kd ? kd_row0 : Bp; a.Np = Np; a.teacher = teacher; a.ldt = ldt; a.trow = trow; a.tlse2 = tlse2;
    FuseArgs128 fa;
    fa.sp_ids = sp_ids; fa.sp_rows = sp_rows; fa.n_sp = n_sp; fa.sp_src = sp_src; fa.sp_scale = sp_scale;
    fa.tg_ids = tg_ids; fa.tg_rows = tg_rows; fa.n_tg = n_tg; fa.wrow = wrow;
    fa.sp_start = sp_start; fa.tg_start = tg_start;
    fa.emb1 = emb + H; fa.m1 = adam_m + H; fa.v1 = adam_v + H; fa.sh1w = (bf16*)shadow + LDR;
    fa.lr_t = lr_t; fa.omb1 = 1.0f - beta1; fa.omb2 = 1.0f - beta2; fa.eps = eps;
    fa.extra1 = extra_grad ? extra_grad + H : nullptr;
    const int all = (N + 127) / 128;
    int tb = tile_begin < 0 ? 0 : tile_begin;
    int te = tile_count < 0 ? all : tb + tile_count;
    if (te > all) te = all;
    if (te <= tb) return 0;
    a.tile_off = tb;
    hipStream_t st = (hipStream_t)stream;
    // 64-row tiles: the tile range (given in 128-row units) in units of 64 rows
    const int all64 = (N + 63) / 64;
    int t0 = 2 * tb, t1 = 2 * te;
    if (t1 > all64) t1 = all64;
    a.tile_off = t0;
    if (kd) hipLaunchKernelGGL((k_tab16<false, true>), dim3(t1 - t0), dim3(256), lds, st, a, fa);
    else if (extra_grad) hipLaunchKernelGGL((k_tab16<true, false>), dim3(t1 - t0), dim3(256), lds, st, a, fa);
    else hipLaunchKernelGGL((k_tab16<false, false>), dim3(t1 - t0), dim3(256), lds, st, a, fa);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_tab_update_sh(const void* rep_bf, void* shadow, int item_num, int B, int Bp, int H, int N, const float* off,
                       const int* sp_ids, const int* sp_rows, const int* sp_start, int n_sp, const float* sp_src, float sp_scale,
                       const int* tg_ids, const int* tg_rows, const int* tg_start, int n_tg, const float* wrow, float* emb,
                       float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                       int tile_count, const float* extra_grad, void* stream) {
    return tab_update_sh(rep_bf, shadow, item_num, B, Bp, H, N, off, sp_ids, sp_rows, sp_start, n_sp, sp_src, sp_scale, tg_ids, tg_rows,
                         tg_start, n_tg, wrow, emb, adam_m, adam_v, lr_t, beta1, beta2, eps, tile_begin, tile_count, extra_grad, Bp, 0,
                         nullptr, 0, nullptr, nullptr, stream);
}

// The same update for a DISTILLED step (ADER.py:132-137): batch rows [kd_row0, Bp) of the padded layout of ader_lbf_fwd_kd are
// exemplar rows whose dlogit is w (softmax(s[:Np]) - softmax(teacher row)); wrow / off / trow / tlse2 as that call left them.
int ader_tab_update_sh_kd(const void* rep_bf, void* shadow, int item_num, int Bp, int kd_row0, int H, int N, int Np, const float* off,
                          const int* sp_ids, const int* sp_rows, const int* sp_start, int n_sp, const float* sp_src, float sp_scale,
                          const int* tg_ids, const int* tg_rows, const int* tg_start, int n_tg, const float* wrow,
                          const float* teacher, long ldt, const int* trow, const float* tlse2, float* emb, float* adam_m,
                          float* adam_v, float lr_t, float beta1, float beta2, float eps, void* stream) {
    return tab_update_sh(rep_bf, shadow, item_num, Bp, Bp, H, N, off, sp_ids, sp_rows, sp_start, n_sp, sp_src, sp_scale, tg_ids, tg_rows,
                         tg_start, n_tg, wrow, emb, adam_m, adam_v, lr_t, beta1, beta2, eps, 0, -1, nullptr, kd_row0, Np, teacher, ldt,
                         trow, tlse2, stream);
}

}  // extern "C"
