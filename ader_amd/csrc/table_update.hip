// Item-table gradient of the full-catalog softmax CE and the fused dense TF-Adam update of the table.
// Reference: ADER.py:91-93 (logits = rep . item_emb^T, one-hot softmax CE), its gradient w.r.t. item_emb, the gradient of
// the input-embedding gather (modules.py:124-130) and tf.train.AdamOptimizer applied densely to the table (ADER.py:96).
//
// One workgroup (4 waves) owns a tile of 64 consecutive table rows for the whole step:
//
//   1. theta tile (64 rows x H fp32, one contiguous block of memory) -> LDS.  It is the ONLY read of the item parameters:
//      the GEMM operand E (bf16, or bf16 hi + lo for the float32-grade "x3" mode) is cut from it into registers, and in
//      bf16 mode the tile stays in LDS until the optimiser phase (no bf16 shadow read, no second theta read).
//   2. dE tile = dlogit^T . rep over all batch rows: per 64-row chunk of rep (staged through LDS) S = rep.E^T is
//      recomputed (v_mfma_f32_32x32x16_bf16, batch rows on the MFMA rows), p = exp2(S*log2e + off_b) is packed to bf16 in
//      registers and fed back as the A operand of dE[item,:] += P^T . rep, rep read k-major with ds_read_b64_tr_b16.
//      Wave (ih, bh) handles items 32*ih.. and the batch rows 32*bh.. of every chunk; the two partial tiles of an item half
//      are summed in LDS in a fixed order.  x3: three MFMAs per product (hi.hi + lo.hi + hi.lo), P split the same way.
//   3. sparse terms from id-sorted lists (input-embedding rows, one-hot targets) added in list order: no atomics,
//      bit-reproducible.
//   4. TF-Adam on the tile, walked as 16-byte vectors of the flat [64*H] block (theta from LDS in bf16 mode), m / v / theta
//      stored once; the bf16 shadow rows of the forward pass are rebuilt from the updated LDS tile in 16-byte pieces.
//
// HBM traffic per table row: theta, m, v in and out (6 x 4H B) + shadow out (336 B) in bf16 mode.  ADAM = false writes the
// dE rows instead (gradient-only entry point for the parity tests and the dense data-parallel exchange).
//
// Which form runs where (round-2 measurements, cfg-S): the x3 mode and the gradient-only entry point use this kernel; the bf16
// fused update uses the 128-row form of table_update_sh.hip, which reads 8 % more bytes (the shadow rows) but is 12 % faster:
// both are bound by the per-workgroup latency chain (two workgroups per CU: registers and LDS allow no more), not by bytes.
// A persistent variant of this kernel (ticketed tiles, next tile's theta / first half of m, v requested across tiles) measured
// 15 % SLOWER: gfx950 counts loads and stores in ONE in-order vmcnt, so the first load a wave waits for in tile t+1 also waits
// for every store of tile t, whereas a workgroup that simply ends never waits for its stores.
#include "lbf_common.h"
#include "../../include/ader_hip.h"

#define TI 64                      // table rows per workgroup
#define PCS_ROW (LDR * 2 / 16)     // 16-byte pieces per bf16 operand row (21)
#define NVEC 10                    // 16-byte vectors per thread covering a tile: 10 * 1024 floats >= 64 * 160 (+ 2 peeled)
#define SPV 3                      // input-embedding gradient rows prefetched under the GEMM phase
#define SPB 4                      // sparse-list entries per batch of the optimiser phase
#define TM_LIST 18                 // ints per list in a tile record: [k0, k1, 8 x (id, row)]

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));


template <bool X3, bool ADAM, bool EXTRA, bool KD = false>
__global__ __launch_bounds__(256, 2) void k_tab_upd(TabArgs a, FuseArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr bool RES = ADAM && !X3;             // theta tile resident in LDS from the operand cut to the optimiser phase
    constexpr int AV = (X3 || EXTRA) ? 5 : 10;    // vectors of a load round (registers: AV x {m, v[, theta][, extra]})
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ih = wave & 1, bh = wave >> 1;
    const int H = a.H, N = a.N;
    const int tile = blockIdx.x + a.tile_off;
    const int tile0 = tile * TI;
    const int rows_avail = min(TI, a.vrows - tile0);
    const int rows_valid = min(TI, N - tile0);
    const int n_av = rows_avail > 0 ? rows_avail * H : 0;
    const int n_el = rows_valid > 0 ? rows_valid * H : 0;
    const float* __restrict__ gsrc = a.emb1 + (size_t)tile0 * H;
    // the tile starts 0 or 8 bytes past a 16-byte boundary (H even): `head` floats are peeled so that vector u of thread t,
    // floats e = head + 4 t + 1024 u, is 16-byte aligned in memory AND in LDS (the LDS images start at the same phase)
    const int ph = (int)(((uintptr_t)gsrc & 15) >> 2);
    const int head = ph ? 4 - ph : 0;
    const int tile_bytes = TI * H * 4 + 16;
    const int work_bytes = max(tile_bytes, TI * LDR * 2 * (X3 ? 2 : 1));
    unsigned char* wk = smem_raw + (RES ? tile_bytes : 0);
    float* T_l = (float*)((RES ? smem_raw : wk) + 4 * ph);      // theta tile, flat [64*H]
    bf16* R_l = (bf16*)wk;                                       // rep chunk: [64][LDR] hi (, [64][LDR] lo)
    float* F_l = (float*)(wk + 4 * ph);                          // dE staging tile, flat [64*H]
    float* off_l = (float*)(wk + work_bytes);                    // [Bp]
    int* meta_l = (int*)(off_l + a.Bp);                          // ADAM: the tile's record [2][TM_LIST] (k_tile_meta)
    float* toff_l = (float*)(meta_l + 2 * TM_LIST);              // KD: [Bp - kd_row0] log2(w_b) - tlse2_b (-inf: no teacher term)
    int* trow_l = (int*)(toff_l + (a.Bp - a.kd_row0));           // KD: [Bp - kd_row0] teacher row (0 for padding rows)

    // ---- theta tile -> LDS (zero beyond the table's last row)
    {
        f32x4_t t4[NVEC];
#pragma unroll
        for (int u = 0; u < NVEC; ++u) {
            const int e = head + 4 * tid + 1024 * u;
            t4[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (e + 3 < n_av) t4[u] = *(const f32x4_t*)(gsrc + e);
            else if (e + 1 < n_av) { const f32x2_t t2 = *(const f32x2_t*)(gsrc + e); t4[u][0] = t2[0]; t4[u][1] = t2[1]; }
        }
        f32x2_t h2 = (f32x2_t){0.f, 0.f};
        if (head && tid == 0 && n_av > 0) h2 = *(const f32x2_t*)gsrc;
        for (int i = tid; i < a.Bp; i += 256) off_l[i] = a.off[i];
        if (KD) {
            for (int i = tid; i < a.Bp - a.kd_row0; i += 256) {
                const int b = a.kd_row0 + i, tr = a.trow[b];
                const float w = f.wrow[b];
                toff_l[i] = (tr >= 0 && w > 0.0f) ? log2f(w) - a.tlse2[b] : -INFINITY;
                trow_l[i] = tr < 0 ? 0 : tr;
            }
        }
        if (ADAM && tid < 2 * TM_LIST) meta_l[tid] = f.tile_meta[(size_t)tile * (2 * TM_LIST) + tid];     // this tile's list record
#pragma unroll
        for (int u = 0; u < NVEC; ++u) {
            const int e = head + 4 * tid + 1024 * u;
            if (e < TI * H) {
                if (e + 3 < TI * H) *(f32x4_t*)(T_l + e) = t4[u];
                else *(f32x2_t*)(T_l + e) = (f32x2_t){t4[u][0], t4[u][1]};
            }
        }
        if (head && tid == 0) *(f32x2_t*)T_l = h2;
    }
    __syncthreads();
    // ADAM: the first input-embedding gradient rows of the tile (thread c holds column c), requested now, used after the GEMM
    float spv[SPV];
    if (ADAM) {
#pragma unroll
        for (int i = 0; i < SPV; ++i)
            spv[i] = (tid < H && meta_l[0] + i < meta_l[1]) ? f.sp_src[(size_t)meta_l[3 + 2 * i] * H + tid] * f.sp_scale : 0.0f;
    }
    // ---- operand fragments: lane (item r of half ih, k-half hh) holds E[item][16 ks + 8 hh + 0..7]
    bf16x8 e_hi[10], e_lo[X3 ? 10 : 1];
    {
        const float* row = T_l + (ih * 32 + r) * H;
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const int col = 16 * ks + 8 * hh + 2 * j2;
                f32x2_t x = (f32x2_t){0.f, 0.f};
                if (col < H) x = *(const f32x2_t*)(row + col);
                const bf16 h0 = (bf16)x[0], h1 = (bf16)x[1];
                e_hi[ks][2 * j2] = h0; e_hi[ks][2 * j2 + 1] = h1;
                if (X3) { e_lo[X3 ? ks : 0][2 * j2] = (bf16)(x[0] - (float)h0); e_lo[X3 ? ks : 0][2 * j2 + 1] = (bf16)(x[1] - (float)h1); }
            }
        }
    }
    const int nch = a.Bp >> 6;
    constexpr int n16 = 64 * LDR * 2 / 16;              // 16-byte pieces per 64-row chunk plane (1344)
    uint4 pf[6];
#define TU_PREFETCH(src_, c_)                                                                            \
    {                                                                                                    \
        const uint4* s_ = (const uint4*)((src_) + (size_t)(c_) * 64 * LDR);                               \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                  \
            const int idx = tid + 256 * j;                                                               \
            pf[j] = (idx < n16) ? s_[idx] : make_uint4(0u, 0u, 0u, 0u);                                   \
        }                                                                                                \
    }
#define TU_STAGE(plane_)                                                                                 \
    {                                                                                                    \
        uint4* d_ = (uint4*)(R_l + (plane_) * 64 * LDR);                                                  \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                  \
            const int idx = tid + 256 * j;                                                               \
            if (idx < n16) d_[idx] = pf[j];                                                              \
        }                                                                                                \
    }
    if (!X3) TU_PREFETCH(a.rep_hi, 0);
    if (!RES) __syncthreads();                          // the operand cut has read T_l, which shares the work area
    if (!X3) { TU_STAGE(0); } else { TU_PREFETCH(a.rep_hi, 0); TU_STAGE(0); TU_PREFETCH(a.rep_lo, 0); TU_STAGE(1); }
    __syncthreads();
    f32x16 dE[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) dE[nb][j] = 0.0f;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const bf16* Rh = R_l + (bh * 32) * LDR;             // this wave's 32 batch rows of the chunk
    const bf16* Rl = Rh + 64 * LDR;
    for (int c = 0; c < nch; ++c) {
        const bool more = c + 1 < nch;
        if (!X3 && more) TU_PREFETCH(a.rep_hi, c + 1);
        const int b0 = c * 64 + bh * 32;
        f32x16 S;
#pragma unroll
        for (int j = 0; j < 16; ++j) S[j] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            const bf16x8 ah = *(const bf16x8*)(Rh + r * LDR + 16 * ks + 8 * hh);
            S = mfma_bf16(ah, e_hi[ks], S);
            if (X3) {
                const bf16x8 al = *(const bf16x8*)(Rl + r * LDR + 16 * ks + 8 * hh);
                S = mfma_bf16(al, e_hi[ks], S);
                S = mfma_bf16(ah, e_lo[X3 ? ks : 0], S);
            }
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
        if (KD && b0 >= a.kd_row0) {   // dlogit of a distilled row: w (softmax(s[:Np]) - softmax(t)) for items < Np, 0 beyond
            const int it = tile0 + ih * 32 + r;                  // (wave-uniform branch: chunks do not straddle kd_row0)
            if (it < a.Np) {
                float tv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) tv[j] = a.teacher[(size_t)trow_l[b0 - a.kd_row0 + acc_row(j, hh)] * a.ldt + it];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    S[j] -= __builtin_amdgcn_exp2f(fmaf(tv[j], LOG2E, toff_l[b0 - a.kd_row0 + acc_row(j, hh)]));
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) S[j] = 0.0f;
            }
        }
        const bf16x8 pa0 = pack8(S, 0), pa1 = pack8(S, 1);
        bf16x8 pl0, pl1;
        if (X3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { pl0[j] = (bf16)(S[j] - (float)pa0[j]); pl1[j] = (bf16)(S[8 + j] - (float)pa1[j]); }
        }
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            const bf16* base = Rh + (4 * hh + q4) * LDR + 32 * nb + 16 * g1 + 4 * p4;
            const bf16x4 l0 = tr_read(base), h0 = tr_read(base + 8 * LDR);
            const bf16x4 l1 = tr_read(base + 16 * LDR), h1 = tr_read(base + 24 * LDR);
            bf16x8 b0v, b1v;
#pragma unroll
            for (int j = 0; j < 4; ++j) { b0v[j] = l0[j]; b0v[4 + j] = h0[j]; b1v[j] = l1[j]; b1v[4 + j] = h1[j]; }
            dE[nb] = mfma_bf16(pa0, b0v, dE[nb]);
            dE[nb] = mfma_bf16(pa1, b1v, dE[nb]);
            if (X3) {
                dE[nb] = mfma_bf16(pl0, b0v, dE[nb]);
                dE[nb] = mfma_bf16(pl1, b1v, dE[nb]);
                const bf16* bl = base + 64 * LDR;
                const bf16x4 m0 = tr_read(bl), n0 = tr_read(bl + 8 * LDR);
                const bf16x4 m1 = tr_read(bl + 16 * LDR), n1 = tr_read(bl + 24 * LDR);
                bf16x8 c0v, c1v;
#pragma unroll
                for (int j = 0; j < 4; ++j) { c0v[j] = m0[j]; c0v[4 + j] = n0[j]; c1v[j] = m1[j]; c1v[4 + j] = n1[j]; }
                dE[nb] = mfma_bf16(pa0, c0v, dE[nb]);
                dE[nb] = mfma_bf16(pa1, c1v, dE[nb]);
            }
        }
        __syncthreads();                                // every wave is done with this chunk
        if (more) {
            if (!X3) { TU_STAGE(0); } else { TU_PREFETCH(a.rep_hi, c + 1); TU_STAGE(0); TU_PREFETCH(a.rep_lo, c + 1); TU_STAGE(1); }
            __syncthreads();
        }
    }
    // ---- optimiser state of the tile: the first load round is requested BEFORE the dE staging and the sparse terms
    // (independent of both); the barriers in between order LDS only, so the loads stay in flight across them
    float* __restrict__ gp = ADAM ? f.emb1 + (size_t)tile0 * H : nullptr;
    float* __restrict__ gm = ADAM ? f.m1 + (size_t)tile0 * H : nullptr;
    float* __restrict__ gv = ADAM ? f.v1 + (size_t)tile0 * H : nullptr;
    const float* __restrict__ gx = EXTRA ? f.extra1 + (size_t)tile0 * H : nullptr;
    f32x4_t P[RES ? 1 : AV], M[AV], V[AV], G[EXTRA ? AV : 1];
    int NV[AV];
#define ROUND_LOAD(u0_)                                                                                    \
    _Pragma("unroll") for (int u = 0; u < AV; ++u) {                                                       \
        const int e = head + 4 * tid + 1024 * ((u0_) + u);                                                 \
        NV[u] = (e + 3 < n_el) ? 2 : ((e + 1 < n_el) ? 1 : 0);                                             \
        if (NV[u] == 2) {                                                                                  \
            M[u] = *(const f32x4_t*)(gm + e); V[u] = *(const f32x4_t*)(gv + e);                            \
            if (!RES) P[u] = *(const f32x4_t*)(gp + e);                                                    \
            if (EXTRA) G[u] = __builtin_nontemporal_load((const f32x4_t*)(gx + e));                        \
        } else if (NV[u] == 1) {                                                                           \
            const f32x2_t m = *(const f32x2_t*)(gm + e), v = *(const f32x2_t*)(gv + e);                    \
            M[u] = (f32x4_t){m[0], m[1], 0.f, 0.f}; V[u] = (f32x4_t){v[0], v[1], 0.f, 0.f};                \
            if (!RES) { const f32x2_t p = *(const f32x2_t*)(gp + e); P[u] = (f32x4_t){p[0], p[1], 0.f, 0.f}; } \
            if (EXTRA) { const f32x2_t g_ = *(const f32x2_t*)(gx + e); G[u] = (f32x4_t){g_[0], g_[1], 0.f, 0.f}; } \
        }                                                                                                  \
    }
    if (ADAM) { ROUND_LOAD(0); }
    // ---- dE accumulators (rows = items, col = channel) -> LDS tile [64][H]; the two batch halves are summed in fixed order
    if (bh == 0) {
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            const int h = 32 * nb + r;
            if (h < H) {
#pragma unroll
                for (int j = 0; j < 16; ++j) F_l[(ih * 32 + acc_row(j, hh)) * H + h] = dE[nb][j];
            }
        }
    }
    lds_only_barrier();
    if (bh == 1) {
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            const int h = 32 * nb + r;
            if (h < H) {
#pragma unroll
                for (int j = 0; j < 16; ++j) F_l[(ih * 32 + acc_row(j, hh)) * H + h] += dE[nb][j];
            }
        }
    }
    lds_only_barrier();
    if (!ADAM) {
        const int HH = H >> 1;
        for (int idx = tid; idx < TI * HH; idx += 256) {
            const int row = idx / HH, c2 = idx - row * HH;
            if (tile0 + row < N)
                *(float2*)(a.demb1 + (size_t)(tile0 + row) * H + 2 * c2) = *(const float2*)(F_l + row * H + 2 * c2);
        }
        return;
    }
    // ---- sparse terms of the tile: item ids [tile0+1, tile0+65).  Thread c owns column c of every row.
    {
        const int id_lo = tile0 + 1, id_hi = min(tile0 + TI, N) + 1;
        if (tid < H && id_lo < id_hi) {
            const int* ms = meta_l;
            const int* mg = meta_l + TM_LIST;
            const int k0s = ms[0], k1s = ms[1];
#pragma unroll
            for (int i = 0; i < SPV; ++i) {                  // rows already in registers (same (id, row) order)
                if (k0s + i < k1s) {
                    const int id = ms[2 + 2 * i];
                    if (id < id_hi) F_l[(id - id_lo) * H + tid] += spv[i];
                }
            }
            // batches of SPB entries: ids and rows, then every gradient row, then the adds in entry order (one dependent memory
            // round trip per entry made a hot item's workgroup the straggler of the launch; see table_update_sh.hip)
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
                for (int u = 0; u < SPB; ++u) {
                    // (unconditional loads -- row 0 for entries that do not count: a load under a per-entry branch is waited for
                    //  at the branch's end, one memory round trip per ENTRY instead of per batch)
                    float rv = (float)a.rep_hi[(size_t)bw[u] * LDR + tid];
                    if (X3) rv += (float)a.rep_lo[(size_t)bw[u] * LDR + tid];
                    val[u] = rv * f.wrow[bw[u]] * ((idv[u] < id_hi) ? 1.0f : 0.0f);
                }
#pragma unroll
                for (int u = 0; u < SPB; ++u)
                    if (idv[u] < id_hi) F_l[(idv[u] - id_lo) * H + tid] -= val[u];
            }
        }
    }
    lds_only_barrier();
    // ---- TF-Adam on the tile (ADER.py:96): m += (g-m)(1-b1); v += (g*g-v)(1-b2); theta -= lr_t*m/(sqrt(v)+eps)
#define ADAM1(p_, m_, v_, g_)                                                                              \
    { m_ += ((g_) - m_) * f.omb1; v_ += ((g_) * (g_) - v_) * f.omb2; p_ -= (m_ * f.lr_t) / (sqrtf(v_) + f.eps); }
    if (head && tid == 0 && n_el > 0) {                   // floats 0,1 (row 0, columns 0,1)
        f32x2_t p = RES ? *(const f32x2_t*)T_l : *(const f32x2_t*)gp;
        f32x2_t m = *(const f32x2_t*)gm, v = *(const f32x2_t*)gv;
        float2 g2 = *(const float2*)F_l;
        if (EXTRA) { g2.x += gx[0]; g2.y += gx[1]; }
        ADAM1(p[0], m[0], v[0], g2.x); ADAM1(p[1], m[1], v[1], g2.y);
        *(f32x2_t*)gp = p; *(f32x2_t*)gm = m; *(f32x2_t*)gv = v;
        if (RES) *(f32x2_t*)T_l = p;
    }
#pragma unroll 1
    for (int u0 = 0; u0 < NVEC; u0 += AV) {
        if (u0) { ROUND_LOAD(u0); }
#pragma unroll
        for (int u = 0; u < AV; ++u) {
            if (NV[u] == 0) continue;
            const int e = head + 4 * tid + 1024 * (u0 + u);
            f32x4_t g4;
            if (NV[u] == 2) g4 = *(const f32x4_t*)(F_l + e);
            else { const f32x2_t g2 = *(const f32x2_t*)(F_l + e); g4 = (f32x4_t){g2[0], g2[1], 0.f, 0.f}; }
            if (EXTRA) g4 += G[u];
            f32x4_t p, m = M[u], v = V[u];
            if (RES) {
                if (NV[u] == 2) p = *(const f32x4_t*)(T_l + e);
                else { const f32x2_t p2 = *(const f32x2_t*)(T_l + e); p = (f32x4_t){p2[0], p2[1], 0.f, 0.f}; }
            } else p = P[RES ? 0 : u];
            ADAM1(p[0], m[0], v[0], g4[0]); ADAM1(p[1], m[1], v[1], g4[1]);
            ADAM1(p[2], m[2], v[2], g4[2]); ADAM1(p[3], m[3], v[3], g4[3]);
            if (NV[u] == 2) {
                // theta/m/v of this tile are not touched again this step: keep them out of the caches
                __builtin_nontemporal_store(p, (f32x4_t*)(gp + e));
                __builtin_nontemporal_store(m, (f32x4_t*)(gm + e));
                __builtin_nontemporal_store(v, (f32x4_t*)(gv + e));
                if (RES) *(f32x4_t*)(T_l + e) = p;
            } else {
                *(f32x2_t*)(gp + e) = (f32x2_t){p[0], p[1]};
                *(f32x2_t*)(gm + e) = (f32x2_t){m[0], m[1]};
                *(f32x2_t*)(gv + e) = (f32x2_t){v[0], v[1]};
                if (RES) *(f32x2_t*)(T_l + e) = (f32x2_t){p[0], p[1]};
            }
        }
    }
#undef ADAM1
    // ---- bf16 shadow rows of the updated tile (operand of the forward logit GEMM), 16-byte pieces, K padding zero
    if (RES && f.sh1w != nullptr) {
        lds_only_barrier();
        bf16* __restrict__ psh = f.sh1w + (size_t)tile0 * LDR;
        const int npc = (rows_valid > 0 ? rows_valid : 0) * PCS_ROW;
        for (int idx = tid; idx < npc; idx += 256) {
            const int row = idx / PCS_ROW, pc = idx - row * PCS_ROW;
            const float* src = T_l + row * H + 8 * pc;
            bf16x8 o;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                f32x2_t x = (f32x2_t){0.f, 0.f};
                if (8 * pc + 2 * j2 < H) x = *(const f32x2_t*)(src + 2 * j2);
                o[2 * j2] = (bf16)x[0]; o[2 * j2 + 1] = (bf16)x[1];
            }
            *(bf16x8*)(psh + (size_t)row * LDR + 8 * pc) = o;
        }
    }
}

// Per-tile records of the two id-sorted sparse lists: rec[tile][list] = {k0, k1, first 8 (id, row) entries of [k0, k1)} where
// [k0, k1) are the entries whose ids fall into the tile's bucket (ids [64 tile + 1, 64 tile + 65)).  One coalesced 144-byte read
// per tile replaces three dependent global round trips inside the update kernel.
__global__ __launch_bounds__(256) void k_tile_meta(const int* __restrict__ sp_ids, const int* __restrict__ sp_rows,
                                                   const int* __restrict__ sp_start, const int* __restrict__ tg_ids,
                                                   const int* __restrict__ tg_rows, const int* __restrict__ tg_start, int ntiles,
                                                   int* __restrict__ rec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * ntiles) return;
    const int tile = i >> 1, lst = i & 1;
    const int* st = lst ? tg_start : sp_start;
    const int* ids = lst ? tg_ids : sp_ids;
    const int* rows = lst ? tg_rows : sp_rows;
    int* mt = rec + (size_t)i * TM_LIST;
    const int k0 = st[tile], k1 = st[tile + 1];
    mt[0] = k0; mt[1] = k1;
    for (int j = 0; j < 8; ++j) {
        const bool in = k0 + j < k1;
        mt[2 + 2 * j] = in ? ids[k0 + j] : 0;
        mt[3 + 2 * j] = in ? rows[k0 + j] : 0;
    }
}

// sparse one-hot term of dlogit for the gradient-only entry point: dE[label_b,:] -= w_b * rep_b  (one wave per batch row).
// Several rows of a batch may share a label: the wave of the FIRST such row (lowest b) is the single writer of that table row and
// subtracts the rows of all of them in batch order; the other waves leave.  No float atomics: the sum has one fixed order
// (deterministic kernels on the parity path).
__global__ __launch_bounds__(256) void k_tab_target_fix(const bf16* __restrict__ rep_hi, const bf16* __restrict__ rep_lo,
                                                        const int* __restrict__ lab, const float* __restrict__ wrow,
                                                        float* __restrict__ demb1, int B, int H) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int lb = lab[b];
    const int t = lb - 1;
    if (t < 0) return;
    // an earlier row with this label?  (64 labels per step; B <= 4096)
    for (int b0 = 0; b0 < b; b0 += 64) {
        const int j = b0 + lane;
        const bool hit = j < b && lab[j] == lb;
        if (__ballot(hit) != 0ull) return;
    }
    float acc[3] = {0.0f, 0.0f, 0.0f};                       // channels lane, lane + 64, lane + 128 (H <= 168)
    for (int b0 = b & ~63; b0 < B; b0 += 64) {
        const int j = b0 + lane;
        unsigned long long m = __ballot(j >= b && j < B && lab[j] == lb);
        while (m) {                                          // rows with this label, ascending
            const int jj = b0 + __builtin_ctzll(m);
            m &= m - 1;
            const float w = wrow[jj];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int c = lane + 64 * r;
                if (c < H) {
                    float rv = (float)rep_hi[(size_t)jj * LDR + c];
                    if (rep_lo) rv += (float)rep_lo[(size_t)jj * LDR + c];
                    acc[r] -= w * rv;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int c = lane + 64 * r;
        if (c < H) demb1[(size_t)t * H + c] += acc[r];
    }
}

// ============================================================================================= C ABI
static size_t tab_lds(int Bp, int H, bool x3, bool adam) {
    const size_t tile = (size_t)TI * H * 4 + 16;
    size_t work = (size_t)TI * LDR * 2 * (x3 ? 2 : 1);
    if (work < tile) work = tile;
    return ((adam && !x3) ? tile : 0) + work + (size_t)Bp * sizeof(float) + 2 * TM_LIST * sizeof(int) + (size_t)Bp * 8;
}

template <bool X3, bool ADAM, bool EXTRA, bool KD = false>
static int tab_launch(const TabArgs& a, const FuseArgs& fa, int tiles, size_t lds, hipStream_t st) {
    static int lds_set_dev[ADER_MAX_DEV] = {};
    int& lds_set = lds_set_dev[ader_cur_dev()];
    if ((int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_tab_upd<X3, ADAM, EXTRA, KD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = (int)lds;
    }
    hipLaunchKernelGGL((k_tab_upd<X3, ADAM, EXTRA, KD>), dim3(tiles), dim3(256), lds, st, a, fa);
    return 0;
}

extern "C" {

// Per-tile records of the id-sorted sparse lists for ader_tab_update: rec [ceil(N/64)][2][18] ints (ader_tab_meta_ints(N)).
// sp_start / tg_start: offsets of the 64-id buckets in the sorted lists (bucket j = ids [64 j + 1, 64 j + 65)), one entry per
// bucket up to and including the bucket that contains N, plus the end offset.
int ader_tab_meta_ints(int N) { return ((N + TI - 1) / TI) * 2 * TM_LIST; }
int ader_tab_tile_meta(const int* sp_ids, const int* sp_rows, const int* sp_start, const int* tg_ids, const int* tg_rows,
                       const int* tg_start, int N, int* rec, void* stream) {
    const int ntiles = (N + TI - 1) / TI;
    if (ntiles <= 0) return 0;
    hipLaunchKernelGGL(k_tile_meta, dim3((2 * ntiles + 255) / 256), dim3(256), 0, (hipStream_t)stream, sp_ids, sp_rows, sp_start,
                       tg_ids, tg_rows, tg_start, ntiles, rec);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Table gradient rows 1..N (overwritten) of the one-hot softmax CE, including the sparse one-hot term.
// rep_hi [Bp,168] bf16 operand rows (ader_lbf_prep); rep_lo: the low-order rows of the x3 mode (ader_lx3_prep) or NULL.
int ader_tab_grad(const void* rep_hi, const void* rep_lo, const float* emb, int item_num, int B, int Bp, int H, int N,
                  const int* lab, const float* wrow, const float* off, float* demb, void* stream) {
    if (B <= 0) return 0;
    if (Bp % 128 != 0 || B > Bp || H > HP || (H & 1) || H < 2 || N > item_num || ((uintptr_t)emb & 7)) return -2;
    hipStream_t st = (hipStream_t)stream;
    TabArgs a;
    a.emb1 = emb + H; a.vrows = item_num; a.rep_hi = (const bf16*)rep_hi; a.rep_lo = (const bf16*)rep_lo; a.off = off;
    a.Bp = Bp; a.H = H; a.N = N; a.tile_off = 0; a.rep_img = nullptr; a.demb1 = demb + H;
    a.kd_row0 = Bp; a.Np = 0; a.teacher = nullptr; a.ldt = 0; a.trow = nullptr; a.tlse2 = nullptr;
    FuseArgs fa = {};
    const int tiles = (N + TI - 1) / TI;
    int rc = rep_lo ? tab_launch<true, false, false>(a, fa, tiles, tab_lds(Bp, H, true, false), st)
                    : tab_launch<false, false, false>(a, fa, tiles, tab_lds(Bp, H, false, false), st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tab_target_fix, dim3((B + 3) / 4), dim3(256), 0, st, (const bf16*)rep_hi, (const bf16*)rep_lo, lab, wrow,
                       demb + H, B, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

// The same for a DISTILLED step (padded row layout of ader_lbf_fwd_kd / ader_lx3_fwd_kd: rows [kd_row0, Bp) are exemplar rows with
// dlogit = w (softmax(s[:Np]) - softmax(teacher row))): the gradient-only form that feeds the dense data-parallel exchange.
int ader_tab_grad_kd(const void* rep_hi, const void* rep_lo, const float* emb, int item_num, int Bp, int kd_row0, int H, int N, int Np,
                     const int* lab, const float* wrow, const float* off, const float* teacher, long ldt, const int* trow,
                     const float* tlse2, float* demb, void* stream) {
    if (Bp <= 0) return 0;
    if (Bp % 128 != 0 || kd_row0 % 128 != 0 || kd_row0 >= Bp || H > HP || (H & 1) || H < 2 || N > item_num || ((uintptr_t)emb & 7) ||
        !teacher || !trow || !tlse2 || Np < 1 || Np > N) return -2;
    hipStream_t st = (hipStream_t)stream;
    TabArgs a;
    a.emb1 = emb + H; a.vrows = item_num; a.rep_hi = (const bf16*)rep_hi; a.rep_lo = (const bf16*)rep_lo; a.off = off;
    a.Bp = Bp; a.H = H; a.N = N; a.tile_off = 0; a.rep_img = nullptr; a.demb1 = demb + H;
    a.kd_row0 = kd_row0; a.Np = Np; a.teacher = teacher; a.ldt = ldt; a.trow = trow; a.tlse2 = tlse2;
    FuseArgs fa = {};
    fa.wrow = wrow;
    const int tiles = (N + TI - 1) / TI;
    int rc = rep_lo ? tab_launch<true, false, false, true>(a, fa, tiles, tab_lds(Bp, H, true, false), st)
                    : tab_launch<false, false, false, true>(a, fa, tiles, tab_lds(Bp, H, false, false), st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tab_target_fix, dim3((Bp + 3) / 4), dim3(256), 0, st, (const bf16*)rep_hi, (const bf16*)rep_lo, lab, wrow,
                       demb + H, Bp, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

#ifdef ADER_XCHECK   // the round-2 fused update (64-row tiles): cross-check kernels of the test build only (libader_xcheck.so)
// Fused: table-gradient GEMM + sparse terms + TF-Adam on table rows 1..N (+ bf16 shadow rows), in one pass.
// sp_ids/sp_rows: the B*T input positions sorted by item id (pads = id 0 first) and their row index into sp_src [B*T,H]
// (the masked/dropout-scaled gradient rows left by ader_embed_bwd_rows); sp_scale = sqrt(H).  tg_ids/tg_rows: the labels
// sorted by id and their batch row.  tile_meta: per-tile list records built by ader_tab_tile_meta from the bucket offsets.
// emb/adam_m/adam_v: fp32 [item_num+1, H], same 16-byte phase.  shadow: bf16 [item_num+1, 168] rewritten for the updated rows
// (NULL: none; always none in x3 mode).  tile_begin/tile_count: range of 128-item tiles (count < 0: all) for row-sharded
// updates.  extra_grad: dense fp32 gradient [item_num+1, H] added row by row before the update, or NULL.
int ader_tab_update(const void* rep_hi, const void* rep_lo, void* shadow, int item_num, int B, int Bp, int H, int N, const float* off,
                    const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src, float sp_scale,
                    const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta, const float* wrow, float* emb,
                    float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                    int tile_count, const float* extra_grad, void* stream) {
    if (B <= 0) return 0;
    if (Bp % 128 != 0 || B > Bp || H > HP || (H & 1) || H < 2 || N > item_num) return -2;
    const uintptr_t ph = (uintptr_t)emb & 15;
    if ((ph & 7) || ((uintptr_t)adam_m & 15) != ph || ((uintptr_t)adam_v & 15) != ph) return -2;
    if (extra_grad && ((uintptr_t)extra_grad & 15) != ph) return -2;
    TabArgs a;
    a.emb1 = emb + H; a.vrows = item_num; a.rep_hi = (const bf16*)rep_hi; a.rep_lo = (const bf16*)rep_lo; a.off = off;
    a.Bp = Bp; a.H = H; a.N = N; a.tile_off = 0; a.rep_img = nullptr; a.demb1 = nullptr;
    a.kd_row0 = Bp; a.Np = 0; a.teacher = nullptr; a.ldt = 0; a.trow = nullptr; a.tlse2 = nullptr;
    FuseArgs fa;
    fa.sp_ids = sp_ids; fa.sp_rows = sp_rows; fa.n_sp = n_sp; fa.sp_src = sp_src; fa.sp_scale = sp_scale;
    fa.tg_ids = tg_ids; fa.tg_rows = tg_rows; fa.n_tg = n_tg; fa.wrow = wrow;
    fa.tile_meta = tile_meta;
    fa.emb1 = emb + H; fa.m1 = adam_m + H; fa.v1 = adam_v + H;
    fa.sh1w = (shadow && !rep_lo) ? (bf16*)shadow + LDR : nullptr;
    fa.lr_t = lr_t; fa.omb1 = 1.0f - beta1; fa.omb2 = 1.0f - beta2; fa.eps = eps;
    fa.extra1 = extra_grad ? extra_grad + H : nullptr;
    // tiles [tile_begin, tile_begin + tile_count) of the ceil(N/128) 128-item tiles (tile_count < 0: all) = two 64-row tiles each
    const int all = (N + TI - 1) / TI;
    int tb = (tile_begin < 0 ? 0 : tile_begin) * 2;
    int te = tile_count < 0 ? all : tb + tile_count * 2;
    if (te > all) te = all;
    if (te <= tb) return 0;
    a.tile_off = tb;
    hipStream_t st = (hipStream_t)stream;
    const bool x3 = rep_lo != nullptr;
    const size_t lds = tab_lds(Bp, H, x3, true);
    int rc;
    if (x3) rc = extra_grad ? tab_launch<true, true, true>(a, fa, te - tb, lds, st) : tab_launch<true, true, false>(a, fa, te - tb, lds, st);
    else rc = extra_grad ? tab_launch<false, true, true>(a, fa, te - tb, lds, st) : tab_launch<false, true, false>(a, fa, te - tb, lds, st);
    if (rc) return rc;
    HIP_LAUNCH_CHECK();
    return 0;
}

// The x3-mode update of a DISTILLED step (ADER.py:132-137): batch rows [kd_row0, Bp) of the padded layout of ader_lx3_fwd_kd are
// exemplar rows whose dlogit is w (softmax(s[:Np]) - softmax(teacher row)); wrow / off / trow / tlse2 as that call left them.
int ader_tab_update_kd(const void* rep_hi, const void* rep_lo, int item_num, int Bp, int kd_row0, int H, int N, int Np, const float* off,
                       const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src, float sp_scale, const int* tg_ids,
                       const int* tg_rows, int n_tg, const int* tile_meta, const float* wrow, const float* teacher, long ldt,
                       const int* trow, const float* tlse2, float* emb, float* adam_m, float* adam_v, float lr_t, float beta1,
                       float beta2, float eps, void* stream) {
    if (Bp <= 0) return 0;
    if (Bp % 128 != 0 || kd_row0 % 128 != 0 || kd_row0 >= Bp || H > HP || (H & 1) || H < 2 || N > item_num || !rep_lo || !teacher ||
        !trow || !tlse2 || Np < 1 || Np > N) return -2;
    const uintptr_t ph = (uintptr_t)emb & 15;
    if ((ph & 7) || ((uintptr_t)adam_m & 15) != ph || ((uintptr_t)adam_v & 15) != ph) return -2;
    TabArgs a;
    a.emb1 = emb + H; a.vrows = item_num; a.rep_hi = (const bf16*)rep_hi; a.rep_lo = (const bf16*)rep_lo; a.off = off;
    a.Bp = Bp; a.H = H; a.N = N; a.tile_off = 0; a.rep_img = nullptr; a.demb1 = nullptr;
    a.kd_row0 = kd_row0; a.Np = Np; a.teacher = teacher; a.ldt = ldt; a.trow = trow; a.tlse2 = tlse2;
    FuseArgs fa;
    fa.sp_ids = sp_ids; fa.sp_rows = sp_rows; fa.n_sp = n_sp; fa.sp_src = sp_src; fa.sp_scale = sp_scale;
    fa.tg_ids = tg_ids; fa.tg_rows = tg_rows; fa.n_tg = n_tg; fa.wrow = wrow;
    fa.tile_meta = tile_meta;
    fa.emb1 = emb + H; fa.m1 = adam_m + H; fa.v1 = adam_v + H; fa.sh1w = nullptr;
    fa.lr_t = lr_t; fa.omb1 = 1.0f - beta1; fa.omb2 = 1.0f - beta2; fa.eps = eps;
    fa.extra1 = nullptr;
    int rc = tab_launch<true, true, false, true>(a, fa, (N + TI - 1) / TI, tab_lds(Bp, H, true, true), (hipStream_t)stream);
    if (rc) return rc;
    HIP_LAUNCH_CHECK();
    return 0;
}

#endif  // ADER_XCHECK

// bucket layout the caller must use for sp_start / tg_start: granularity (ids per bucket) and first id of bucket 0
// (a tile of the update covers item ids [64j + 1, 64j + 65))
int ader_fused_bucket_gran(void) { return TI; }
int ader_fused_bucket_id0(void) { return 1; }

}  // extern "C"
