// Fused table update at float32 grade (x3 mode): table-gradient GEMM + sparse terms + TF-Adam (ADER.py:96) in one pass over the
// fp32 table, every product as three bf16 MFMAs on hi/lo operand splits (hi.hi + lo.hi + hi.lo, ~2^-16 relative, fp32 accumulate).
// Reference: ADER.py:91-93 (logits = rep . item_emb^T, softmax CE), its gradient w.r.t. item_emb, the gradient of the input
// gather (modules.py:124-130) and tf.train.AdamOptimizer applied densely to the table (ADER.py:96).
//
// Shape of k_tab16 (table_update_sh.hip): 64-row tiles on v_mfma_f32_16x16x32_bf16, each of the 4 waves owns 16 table rows for the
// WHOLE batch (no cross-wave reduction), <= 168 registers and ~45 KB of LDS so that THREE workgroups share a CU -- per tile the
// matrix pipe needs 15,360 clocks and the theta/m/v stream ~25,000 clocks of the CU's HBM share, and it is the co-resident
// workgroups in different phases that overlap the two.  What differs from the bf16 kernel:
//   * there is no bf16 shadow in x3 mode: the E operand (hi and lo fragments, 40 registers) is cut from the fp32 theta tile, which
//     passes through LDS once at the start of the tile;
//   * rep (hi and lo planes) is streamed in chunks of 32 batch rows by LDS-DMA (global_load_lds_dwordx4, no staging registers)
//     into a double buffer: the chunk after the current one is in flight under the current chunk's 60 MFMAs, one workgroup barrier
//     per chunk (the round-2 kernel loaded, staged and waited for both planes of every chunk synchronously: 1.43-1.53 ms per 10^6
//     rows).  The DMA copies a ready-made LDS IMAGE of the chunk (k_x3_rep_image): 16-byte k-chunks [kc][row][8 elements] placed
//     so that BOTH operand reads are free of bank conflicts -- with the row-major 336-byte rows of the other kernels the
//     ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads of the 16x16x32 operand maps are two-way conflicts
//     (SQ_LDS_BANK_CONFLICT 30 % of the kernel, the LDS pipe busier than the matrix pipe: 1.16 ms);
//   * 32-row chunks: one K = 32 MFMA per (16-channel block, term) -- the A fragment of lane (item c16, k-group g) is its own p
//     values of the chunk's two S blocks (rows 4g..4g+3 of each), the B fragment reads exactly those rep rows k-major with
//     ds_read_b64_tr_b16.
// The optimiser phase (dE staging tile, sparse terms from the bucketed lists in list order -- no atomics, bit-reproducible --, TF-Adam
// over the tile's flat [64*H] block of theta/m/v in 16-byte vectors) is the one of k_tab16 without the shadow rows.  gfx950 only.
#include <stdlib.h>
#include "lbf_common.h"
#include "x3_image.h"
#include "../../include/ader_hip.h"

#define TI 64                      // table rows per workgroup
#define X3_CH 32                   // batch rows per rep chunk
#define TM_LIST 18                 // ints per list in a tile record: [k0, k1, 8 x (id, row)]
#define NVEC 10                    // 16-byte vectors per thread covering a tile: 10 * 1024 floats >= 64 * 160

#define SPV 3                      // input-embedding gradient rows prefetched under the GEMM phase
#define HEAVY_N 32                 // a bucket with more entries than this in either list takes the heavy path
#define HVB 16                     // gradient rows in flight per thread on the heavy path
#define SPB 8                      // sparse-list entries per batch of the optimiser phase (loads of a batch are independent)
#ifndef AV
#define AV 6                       // 16-byte vectors per thread and load round of the optimiser phase (x theta, m, v)
#endif

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ int x3_cu_arrivals[4096];

// One chunk image (22 KiB, contiguous in memory) -> LDS buffer, as 22 LDS-DMA pieces of 1 KiB dealt round-robin to the 4 waves.
// The DMA is issued from inline asm ON PURPOSE: hipcc counts a __builtin_amdgcn_global_load_lds as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of the next ds_read_b64_tr_b16 it cannot tell apart from the destination -- in the middle of the
// chunk, which drained the prefetch half a chunk after it was issued.  Hidden from the compiler, the pieces are waited for by the
// explicit vmcnt(0) at the head of the next chunk only (cdna_hip_programming.md 5.7: M0 saved, written and restored in ONE statement).
__device__ __forceinline__ void x3_glds16(const char* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void x3_dma_chunk(const char* __restrict__ img, int c, bf16* buf, int wave_u, int lane) {
    const char* src0 = img + (size_t)c * X3_IMG_B + 16 * lane;
    const unsigned dst0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)buf;
#pragma unroll
    for (int i = 0; i < (X3_PIECES + 3) / 4; ++i) {
        const int p = wave_u + 4 * i;                       // wave-uniform
        if (p < X3_PIECES) x3_glds16(src0 + 1024 * p, __builtin_amdgcn_readfirstlane(dst0 + 1024 * p));
    }
}

// LDS image of rep for k_tab16x3: img [Bp / 32 chunks][X3_IMG_B]; thread = one 16-byte slot (plane, k-chunk, row) of a chunk
__global__ __launch_bounds__(256) void k_x3_rep_image(const bf16* __restrict__ rep_hi, const bf16* __restrict__ rep_lo, int Bp,
                                                      char* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 2 * 20 * X3_CH;
    if (i >= (Bp / X3_CH) * per) return;
    const int c = i / per, s_ = i - c * per;
    const int plane = s_ / (20 * X3_CH), t = s_ - plane * (20 * X3_CH);
    const int row = t / 20, kc = t - row * 20;              // consecutive threads: consecutive 16-byte pieces of a source row
    const bf16* src = (plane ? rep_lo : rep_hi) + (size_t)(c * X3_CH + row) * LDR + 8 * kc;
    *(uint4*)(img + (size_t)c * X3_IMG_B + plane * X3_PLANE_B + x3_kc_off(kc) + 16 * row) = *(const uint4*)src;
}

// one round of theta / m / v vectors of the tile's flat [64*H] block: all loads issued before any math or store
#define ROUND_LOAD()                                                                                       \
_Pragma("unroll") for (int u = 0; u < AV; ++u) {                                                           \
    E[u] = e;                                                                                              \
    NV[u] = (e + 3 < n_el) ? 2 : ((e + 1 < n_el) ? 1 : 0);                                                 \
    if (NV[u] == 2) {                                                                                      \
        P[u] = *(const f32x4_t*)(gp + e); M[u] = *(const f32x4_t*)(gm + e); V[u] = *(const f32x4_t*)(gv + e); \
        if (EXTRA) G[u] = __builtin_nontemporal_load((const f32x4_t*)(gx + e));                            \
    } else if (NV[u] == 1) {                                                                               \
        if (EXTRA) { const f32x2_t g_ = *(const f32x2_t*)(gx + e); G[u] = (f32x4_t){g_[0], g_[1], 0.f, 0.f}; } \
        const f32x2_t p = *(const f32x2_t*)(gp + e), m = *(const f32x2_t*)(gm + e), v = *(const f32x2_t*)(gv + e); \
        P[u] = (f32x4_t){p[0], p[1], 0.f, 0.f}; M[u] = (f32x4_t){m[0], m[1], 0.f, 0.f}; V[u] = (f32x4_t){v[0], v[1], 0.f, 0.f}; \
    }                                                                                                      \
    e += 1024;                                                                                             \
}

template <bool EXTRA, bool KD>
__global__ __launch_bounds__(256, 3) void k_tab16x3(TabArgs a, FuseArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R_l = (bf16*)smem_raw;                                // [2 buffers][chunk image]; first the theta tile, last the dE tile
    float* off_l = (float*)(smem_raw + 2 * X3_IMG_B);           // [Bp]
    int* meta_l = (int*)(off_l + a.Bp);                         // the tile's list record [2][TM_LIST] (ader_tab_tile_meta)
    float* toff_l = (float*)(meta_l + 2 * TM_LIST);             // KD: [Bp - kd_row0] log2(w_b) - tlse2_b (-inf: no teacher term)
    int* trow_l = (int*)(toff_l + (a.Bp - a.kd_row0));          // KD: [Bp - kd_row0] teacher row (0 for padding rows)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int H = a.H, N = a.N;
    const int tile = blockIdx.x + a.tile_off;
    const int tile0 = tile * TI;
    const int it0 = tile0 + wave * 16;
    const int rows_avail = min(TI, a.vrows - tile0);
    const int n_av = rows_avail > 0 ? rows_avail * H : 0;
    const float* __restrict__ gsrc = a.emb1 + (size_t)tile0 * H;
    // the tile starts 0 or 8 bytes past a 16-byte boundary (H even): `head` floats are peeled so that vector u of thread t,
    // floats e = head + 4 t + 1024 u, is 16-byte aligned in memory AND in LDS (the LDS image starts at the same phase)
    const int ph = (int)(((uintptr_t)gsrc & 15) >> 2);
    const int head = ph ? 4 - ph : 0;
    float* T_l = (float*)(smem_raw + 4 * ph);                   // theta tile, flat [64*H]
    // ---- phase stagger of the first generation of workgroups (speed only): all tiles cost the same, so the three
    // workgroups of a CU would run their matrix phases together and their HBM phases together for the whole launch
    if ((a.ko >> 16) && blockIdx.x < 768) {
        const int mode = (a.ko >> 8) & 15;
        int k;
        if (mode == 3) {        // arrival order on this CU (HW_ID: cu 11:8, sh 12, se 15:13; XCC_ID 3:0)
            int kk = 0;
            if (tid == 0) {
                const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
                const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID
                const unsigned cu = ((hw >> 8) & 0xff) | ((xcc & 15) << 8);
                kk = atomicAdd(&x3_cu_arrivals[cu & 4095], 1) % 3;
            }
            k = __shfl(kk, 0, 64);
            k = __builtin_amdgcn_readfirstlane(k);
            __shared__ int k_sh;
            if (tid == 0) k_sh = k;
            __syncthreads();
            k = k_sh;
        } else k = mode == 0 ? (int)(blockIdx.x >> 8) : (mode == 1 ? (int)((blockIdx.x >> 3) % 3) : (int)(blockIdx.x % 3));
        if (k) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            const unsigned long long dt = (unsigned long long)k * (a.ko >> 16) * 100ull;      // 100 MHz ticks
            while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(64);
        }
    }
    // ---- theta tile -> LDS (zero beyond the table's last row); the per-row constants and the tile's list record beside it
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
        if (tid < 2 * TM_LIST) meta_l[tid] = f.tile_meta[(size_t)tile * (2 * TM_LIST) + tid];
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
    // the first input-embedding gradient rows of the tile (thread c holds column c), requested now, used after the GEMM
    float spv[SPV];
#pragma unroll
    for (int i = 0; i < SPV; ++i)
        spv[i] = (tid < H && meta_l[0] + i < meta_l[1]) ? f.sp_src[(size_t)meta_l[3 + 2 * i] * H + tid] * f.sp_scale : 0.0f;
    // ---- operand fragments: lane (item c16 of this wave's 16, k-group g) holds E[item][32 ks + 8 g + 0..7] as hi + lo
    bf16x8 e_hi[5], e_lo[5];
    {
        const float* row = T_l + (wave * 16 + c16) * H;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const int col = 32 * ks + 8 * g + 2 * j2;
                f32x2_t x = (f32x2_t){0.f, 0.f};
                if (col < H) x = *(const f32x2_t*)(row + col);
                const bf16 h0 = (bf16)x[0], h1 = (bf16)x[1];
                e_hi[ks][2 * j2] = h0; e_hi[ks][2 * j2 + 1] = h1;
                e_lo[ks][2 * j2] = (bf16)(x[0] - (float)h0); e_lo[ks][2 * j2 + 1] = (bf16)(x[1] - (float)h1);
            }
        }
    }
    __syncthreads();                                            // every wave has cut its fragments: the area is free for rep
    f32x4v dE[10];
#pragma unroll
    for (int cb = 0; cb < 10; ++cb) dE[cb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const int nch = a.Bp / X3_CH;
    const int q4 = c16 >> 2, p4 = c16 & 3;
    const char* img = (const char*)a.rep_img;
    // per-lane byte offsets into a chunk image: row read of (row c16, k-group g); transposed read of (row 4g + q4, 4 channels p4)
    const int a_off = 1152 * (g >> 1) + 512 * (g & 1) + 16 * c16;
    const int t_off = 1152 * (p4 >> 1) + 16 * (4 * g + q4) + 8 * (p4 & 1);
    x3_dma_chunk(img, 0, R_l, wave, lane);
    for (int c = 0; c < ((a.ko & 2) ? 0 : nch); ++c) {
        // this wave's pieces of chunk c have landed and its LDS reads of chunk c-1 are done; after the barrier that holds for every
        // wave, so chunk c can be read and the other buffer (chunk c-1's) can be refilled
        if (!(a.ko & 8)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + 1 < nch && !(a.ko & 4)) x3_dma_chunk(img, c + 1, R_l + ((c + 1) & 1) * X3_BUF, wave, lane);
        const char* Bh = (const char*)(R_l + (c & 1) * X3_BUF);
        const int b0 = c * X3_CH;
        // KD rows: this lane's 8 teacher logits (item it0 + c16, batch rows b0 + 16 rb + 4 g + j), requested ahead of the MFMAs
        float tv[KD ? 8 : 1];
        const bool kdc = KD && b0 >= a.kd_row0;                 // (workgroup-uniform: chunks do not straddle kd_row0)
        if (kdc && it0 + c16 < a.Np) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                tv[KD ? j : 0] = a.teacher[(size_t)trow_l[b0 - a.kd_row0 + 16 * (j >> 2) + 4 * g + (j & 3)] * a.ldt + it0 + c16];
        }
        // S block rb = 16 batch rows x 16 items: A = rep rows (lane: row c16 of the block, k = 8g..8g+7), B = this wave's E fragments.
        // The operand reads are software-pipelined by hand, two k-steps ahead of the MFMAs that consume them (left to itself hipcc
        // reloads ONE register set just before its use and every pair of MFMAs waits out a full LDS latency: each wave ran its
        // chunk at a quarter of the matrix rate and a workgroup in its optimiser phase took its share of the pipe with it)
#define X3_LOADA(set_, ks_)                                                                               \
        { const char* ap_ = Bh + a_off + X3_QUAD * (ks_);     /* k-chunk 4 ks + g, row c16 (S block 1: + 16 rows = 256 B) */ \
          set_[0] = *(const bf16x8*)ap_; set_[1] = *(const bf16x8*)(ap_ + X3_PLANE_B);                    \
          set_[2] = *(const bf16x8*)(ap_ + 256); set_[3] = *(const bf16x8*)(ap_ + X3_PLANE_B + 256); }
        // transposed reads of channel block cb = (quad Q = cb >> 1, o = cb & 1): k-chunks 4Q + o and 4Q + o + 2, i.e. lane c16 <->
        // channel 32 Q + 8 o + 16 (c16 >> 3) + (c16 & 7); set = {hi rows 4g.., hi rows 16+4g.., lo rows 4g.., lo rows 16+4g..}
#define X3_LOADT(set_, cb_)                                                                               \
        { const bf16* tp_ = (const bf16*)(Bh + t_off + X3_QUAD * ((cb_) >> 1) + 512 * ((cb_) & 1));         \
          set_[0] = tr_read(tp_); set_[1] = tr_read(tp_ + 128);                                           \
          set_[2] = tr_read(tp_ + X3_PLANE_B / 2); set_[3] = tr_read(tp_ + X3_PLANE_B / 2 + 128); }
        f32x4v S0 = (f32x4v){0.f, 0.f, 0.f, 0.f}, S1 = (f32x4v){0.f, 0.f, 0.f, 0.f};
        bf16x8 fa[2][4];
        X3_LOADA(fa[0], 0);
        X3_LOADA(fa[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            bf16x8* A_ = fa[ks & 1];                             // {ah0, al0, ah1, al1}
            S0 = mfma16_bf16(A_[1], e_hi[ks], S0);
            S1 = mfma16_bf16(A_[3], e_hi[ks], S1);
            S0 = mfma16_bf16(A_[0], e_lo[ks], S0);
            S1 = mfma16_bf16(A_[2], e_lo[ks], S1);
            S0 = mfma16_bf16(A_[0], e_hi[ks], S0);
            S1 = mfma16_bf16(A_[2], e_hi[ks], S1);
            if (ks + 2 < 5) X3_LOADA(fa[ks & 1], ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the first transposed reads of the P^T.rep phase do not depend on S: in flight under the exp2 section
        bf16x4 ft[3][4];
        X3_LOADT(ft[0], 0);
        X3_LOADT(ft[1], 1);
        X3_LOADT(ft[2], 2);
        __builtin_amdgcn_sched_barrier(0);
        // rows of S are batch rows: p = w_b * softmax = exp2(S*log2e + off_b)
        {
            const float4 o0 = *(const float4*)(off_l + b0 + 4 * g);
            const float4 o1 = *(const float4*)(off_l + b0 + 16 + 4 * g);
            S0[0] = __builtin_amdgcn_exp2f(fmaf(S0[0], LOG2E, o0.x)); S0[1] = __builtin_amdgcn_exp2f(fmaf(S0[1], LOG2E, o0.y));
            S0[2] = __builtin_amdgcn_exp2f(fmaf(S0[2], LOG2E, o0.z)); S0[3] = __builtin_amdgcn_exp2f(fmaf(S0[3], LOG2E, o0.w));
            S1[0] = __builtin_amdgcn_exp2f(fmaf(S1[0], LOG2E, o1.x)); S1[1] = __builtin_amdgcn_exp2f(fmaf(S1[1], LOG2E, o1.y));
            S1[2] = __builtin_amdgcn_exp2f(fmaf(S1[2], LOG2E, o1.z)); S1[3] = __builtin_amdgcn_exp2f(fmaf(S1[3], LOG2E, o1.w));
        }
        if (kdc) {              // dlogit of a distilled row: w (softmax(s[:Np]) - softmax(t)) for items < Np, 0 beyond
            if (it0 + c16 < a.Np) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    S0[j] -= __builtin_amdgcn_exp2f(fmaf(tv[KD ? j : 0], LOG2E, toff_l[b0 - a.kd_row0 + 4 * g + j]));
                    S1[j] -= __builtin_amdgcn_exp2f(fmaf(tv[KD ? 4 + j : 0], LOG2E, toff_l[b0 - a.kd_row0 + 16 + 4 * g + j]));
                }
            } else {
                S0 = (f32x4v){0.f, 0.f, 0.f, 0.f}; S1 = (f32x4v){0.f, 0.f, 0.f, 0.f};
            }
        }
        bf16x8 ph_, pl_;        // k order of the fragment: rows 4g..4g+3 of S block 0, then of S block 1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16 h0 = (bf16)S0[j], h1 = (bf16)S1[j];
            ph_[j] = h0; ph_[4 + j] = h1;
            pl_[j] = (bf16)(S0[j] - (float)h0); pl_[4 + j] = (bf16)(S1[j] - (float)h1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < 10; ++cb) {
            bf16x4* T_ = ft[cb % 3];
            bf16x8 bh, bl;
#pragma unroll
            for (int j = 0; j < 4; ++j) { bh[j] = T_[0][j]; bh[4 + j] = T_[1][j]; bl[j] = T_[2][j]; bl[4 + j] = T_[3][j]; }
            dE[cb] = mfma16_bf16(pl_, bh, dE[cb]);
            dE[cb] = mfma16_bf16(ph_, bl, dE[cb]);
            dE[cb] = mfma16_bf16(ph_, bh, dE[cb]);
            if (cb + 3 < 10) X3_LOADT(ft[cb % 3], cb + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef X3_LOADA
#undef X3_LOADT
    // ---- optimiser phase: the tile's rows are ONE contiguous block of 64*H floats in theta / m / v (and in F_l)
    float* F_l = (float*)smem_raw;
    const int rows_valid = min(TI, N - tile0);
    const int n_el = (rows_valid > 0 && !(a.ko & 1)) ? rows_valid * H : 0;
    const size_t tq = (a.ko & 16) ? (size_t)(tile & 63) * TI : (size_t)tile0;    // ko 16: optimiser traffic served by L2 (timing only)
    float* __restrict__ gp = f.emb1 + tq * H;
    float* __restrict__ gm = f.m1 + tq * H;
    float* __restrict__ gv = f.v1 + tq * H;
    const int head2 = (((uintptr_t)gp) & 15) ? 2 : 0;
    int e = head2 + 4 * tid;
    f32x4_t P[AV], M[AV], V[AV], G[EXTRA ? AV : 1];
    const float* __restrict__ gx = EXTRA ? f.extra1 + (size_t)tile0 * H : nullptr;
    int E[AV], NV[AV];
    // A bucket that holds a hot item (Zipf ids: hundreds of entries) takes the HEAVY path below: its (id, row) lists are fetched
    // cooperatively, 256 entries per round trip, and the gradient rows HVB at a time -- with the optimiser loads requested AFTER
    // the sparse terms, so that the registers are free for the deeper batches (a workgroup-uniform choice; rare tiles).
    const bool heavy = (meta_l[1] - meta_l[0] > HEAVY_N) || (meta_l[TM_LIST + 1] - meta_l[TM_LIST] > HEAVY_N);
    if (!heavy) { ROUND_LOAD(); }   // first round of theta/m/v: requested BEFORE the dE staging and the sparse terms
    lds_only_barrier();             // every wave is done with the last rep chunk
#pragma unroll
    for (int cb = 0; cb < 10; ++cb) {
        const int h = 32 * (cb >> 1) + 8 * (cb & 1) + 16 * (c16 >> 3) + (c16 & 7);
        if (h < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) F_l[(wave * 16 + 4 * g + j) * H + h] = dE[cb][j];
        }
    }
    lds_only_barrier();
    {
        // sparse terms of the tile: item ids [tile0+1, tile0+65).  Thread c owns column c of every row.
        const int id_lo = tile0 + 1, id_hi = min(tile0 + TI, N) + 1;
        if (heavy) {
            int* hv_l = (int*)(smem_raw + TI * HP * sizeof(float));     // [2][256] (id, row) of the current chunk, behind F_l
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
            HEAVY_LIST(meta_l[TM_LIST], meta_l[TM_LIST + 1], f.tg_ids, f.tg_rows,
                       f.wrow[rw] * ((float)a.rep_hi[(size_t)rw * LDR + tid] + (float)a.rep_lo[(size_t)rw * LDR + tid]), -=)
#undef HEAVY_LIST
        } else if (tid < H && id_lo < id_hi) {
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
            // entries SPV..7 of the bucket are in the LDS record, the rest in the global lists.  Batches of SPB entries: ids and
            // rows first, then every gradient row, then the adds in entry order (the order fixes the rounding).  The loads are
            // UNCONDITIONAL (row 0 for entries that do not count): under a per-entry branch hipcc waits for each load at the end of
            // its branch -- one memory round trip per entry
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
                    const float rv = (float)a.rep_hi[(size_t)bw[u] * LDR + tid] + (float)a.rep_lo[(size_t)bw[u] * LDR + tid];
                    val[u] = rv * f.wrow[bw[u]] * ((idv[u] < id_hi) ? 1.0f : 0.0f);
                }
#pragma unroll
                for (int u = 0; u < SPB; ++u)
                    if (idv[u] < id_hi) F_l[(idv[u] - id_lo) * H + tid] -= val[u];
            }
        }
    }
    if (heavy) { ROUND_LOAD(); }
    lds_only_barrier();
#define ADAM1(p_, m_, v_, g_)                                                                              \
    { m_ += ((g_) - m_) * f.omb1; v_ += ((g_) * (g_) - v_) * f.omb2; p_ -= (m_ * f.lr_t) / (sqrtf(v_) + f.eps); }
    if (head2 && tid == 0 && n_el > 0) {                  // elements 0,1 (row 0, columns 0,1)
        f32x2_t p = *(const f32x2_t*)gp, m = *(const f32x2_t*)gm, v = *(const f32x2_t*)gv;
        float2 g2 = *(const float2*)F_l;
        if (EXTRA) { g2.x += gx[0]; g2.y += gx[1]; }
        ADAM1(p[0], m[0], v[0], g2.x); ADAM1(p[1], m[1], v[1], g2.y);
        *(f32x2_t*)gp = p; *(f32x2_t*)gm = m; *(f32x2_t*)gv = v;
    }
#pragma unroll 1
    for (int k0 = 0; k0 < NVEC; k0 += AV) {               // 10 * 1024 floats >= 64 * 150; round 0 is already in flight
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
            if (NV[u] == 2) {
                // theta/m/v of this tile are not touched again this step: keep them out of the caches
                __builtin_nontemporal_store(p, (f32x4_t*)(gp + E[u]));
                __builtin_nontemporal_store(m, (f32x4_t*)(gm + E[u]));
                __builtin_nontemporal_store(v, (f32x4_t*)(gv + E[u]));
            } else {
                *(f32x2_t*)(gp + E[u]) = (f32x2_t){p[0], p[1]};
                *(f32x2_t*)(gm + E[u]) = (f32x2_t){m[0], m[1]};
                *(f32x2_t*)(gv + E[u]) = (f32x2_t){v[0], v[1]};
            }
        }
    }
#undef ADAM1
}

// ============================================================================================= launch (C ABI: table_update.hip)
static size_t tab16x3_lds(int Bp, int Bk) {
    return (size_t)2 * X3_IMG_B + (size_t)Bp * sizeof(float) + 2 * TM_LIST * sizeof(int) + (size_t)Bk * 8;
}

template <bool EXTRA, bool KD>
static int tab16x3_launch_t(const TabArgs& a, const FuseArgs& fa, int tiles, size_t lds, hipStream_t st) {
    static int lds_set = 0;
    if ((int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_tab16x3<EXTRA, KD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = (int)lds;
    }
    hipLaunchKernelGGL((k_tab16x3<EXTRA, KD>), dim3(tiles), dim3(256), lds, st, a, fa);
    return 0;
}

static int tab16x3_launch(TabArgs a, const FuseArgs& fa, int tiles, bool extra, bool kd, void* stream) {
    { static int ko = -1; if (ko < 0) { const char* e = getenv("ADER_X3_KO"); ko = e ? atoi(e) : 0; const char* s_ = getenv("ADER_X3_STAGGER"); ko |= (s_ ? atoi(s_) : 0) << 16; } a.ko = ko; }
    if (a.Bp % X3_CH != 0 || (kd && a.kd_row0 % X3_CH != 0) || !a.rep_img || ((uintptr_t)a.rep_img & 15)) return -2;
    const size_t lds = tab16x3_lds(a.Bp, kd ? a.Bp - a.kd_row0 : 0);
    hipStream_t st = (hipStream_t)stream;
    if (kd) return tab16x3_launch_t<false, true>(a, fa, tiles, lds, st);
    if (extra) return tab16x3_launch_t<true, false>(a, fa, tiles, lds, st);
    return tab16x3_launch_t<false, false>(a, fa, tiles, lds, st);
}

extern "C" {

// LDS image of the x3 operand rows for ader_tab_update_x3[_kd]: img = ader_x3_rep_image_bytes(Bp) bytes, 16-byte aligned, built from
// the two planes rep_hi / rep_lo [Bp,168] that ader_lx3_prep / ader_lx3_fwd[_kd] leave (Bp % 32 == 0).
int ader_x3_rep_image_bytes(int Bp) { return (Bp / X3_CH) * X3_IMG_B; }
int ader_x3_rep_image(const void* rep_hi, const void* rep_lo, int Bp, void* img, void* stream) {
    if (Bp <= 0) return 0;
    if (Bp % X3_CH != 0 || ((uintptr_t)img & 15)) return -2;
    const int n = (Bp / X3_CH) * 2 * 20 * X3_CH;
    hipLaunchKernelGGL(k_x3_rep_image, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)rep_hi,
                       (const bf16*)rep_lo, Bp, (char*)img);
    HIP_LAUNCH_CHECK();
    return 0;
}


// The fused table update at float32 grade (gradient GEMM + sparse rows + TF-Adam on table rows 1..N in one pass, ADER.py:91-96):
// arguments as ader_tab_update with rep_lo != NULL, plus rep_img = ader_x3_rep_image of the same two planes; no shadow.
int ader_tab_update_x3(const void* rep_hi, const void* rep_lo, const void* rep_img, int item_num, int B, int Bp, int H, int N,
                       const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src, float sp_scale,
                       const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta, const float* wrow, float* emb,
                       float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                       int tile_count, const float* extra_grad, void* stream) {
    if (B <= 0) return 0;
    if (Bp % 128 != 0 || B > Bp || H > HP || (H & 1) || H < 2 || N > item_num || !rep_lo) return -2;
    const uintptr_t ph = (uintptr_t)emb & 15;
    if ((ph & 7) || ((uintptr_t)adam_m & 15) != ph || ((uintptr_t)adam_v & 15) != ph) return -2;
    if (extra_grad && ((uintptr_t)extra_grad & 15) != ph) return -2;
    TabArgs a;
    a.emb1 = emb + H; a.vrows = item_num; a.rep_hi = (const bf16*)rep_hi; a.rep_lo = (const bf16*)rep_lo; a.rep_img = rep_img;
    a.off = off; a.Bp = Bp; a.H = H; a.N = N; a.tile_off = 0; a.demb1 = nullptr;
    a.kd_row0 = Bp; a.Np = 0; a.teacher = nullptr; a.ldt = 0; a.trow = nullptr; a.tlse2 = nullptr;
    FuseArgs fa;
    fa.sp_ids = sp_ids; fa.sp_rows = sp_rows; fa.n_sp = n_sp; fa.sp_src = sp_src; fa.sp_scale = sp_scale;
    fa.tg_ids = tg_ids; fa.tg_rows = tg_rows; fa.n_tg = n_tg; fa.wrow = wrow;
    fa.tile_meta = tile_meta;
    fa.emb1 = emb + H; fa.m1 = adam_m + H; fa.v1 = adam_v + H; fa.sh1w = nullptr;
    fa.lr_t = lr_t; fa.omb1 = 1.0f - beta1; fa.omb2 = 1.0f - beta2; fa.eps = eps;
    fa.extra1 = extra_grad ? extra_grad + H : nullptr;
    // tiles [tile_begin, tile_begin + tile_count) of the ceil(N/128) 128-item tiles (tile_count < 0: all) = two 64-row tiles each
    const int all = (N + TI - 1) / TI;
    int tb = (tile_begin < 0 ? 0 : tile_begin) * 2;
    int te = tile_count < 0 ? all : tb + tile_count * 2;
    if (te > all) te = all;
    if (te <= tb) return 0;
    a.tile_off = tb;
    int rc = tab16x3_launch(a, fa, te - tb, extra_grad != nullptr, false, stream);
    if (rc) return rc;
    HIP_LAUNCH_CHECK();
    return 0;
}

// ... and for a DISTILLED step (ADER.py:132-137): arguments as ader_tab_update_kd plus rep_img.
int ader_tab_update_x3_kd(const void* rep_hi, const void* rep_lo, const void* rep_img, int item_num, int Bp, int kd_row0, int H, int N,
                          int Np, const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src,
                          float sp_scale, const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta, const float* wrow,
                          const float* teacher, long ldt, const int* trow, const float* tlse2, float* emb, float* adam_m,
                          float* adam_v, float lr_t, float beta1, float beta2, float eps, void* stream) {
    if (Bp <= 0) return 0;
    if (Bp % 128 != 0 || kd_row0 % 128 != 0 || kd_row0 >= Bp || H > HP || (H & 1) || H < 2 || N > item_num || !rep_lo || !teacher ||
        !trow || !tlse2 || Np < 1 || Np > N) return -2;
    const uintptr_t ph = (uintptr_t)emb & 15;
    if ((ph & 7) || ((uintptr_t)adam_m & 15) != ph || ((uintptr_t)adam_v & 15) != ph) return -2;
    TabArgs a;
    a.emb1 = emb + H; a.vrows = item_num; a.rep_hi = (const bf16*)rep_hi; a.rep_lo = (const bf16*)rep_lo; a.rep_img = rep_img;
    a.off = off; a.Bp = Bp; a.H = H; a.N = N; a.tile_off = 0; a.demb1 = nullptr;
    a.kd_row0 = kd_row0; a.Np = Np; a.teacher = teacher; a.ldt = ldt; a.trow = trow; a.tlse2 = tlse2;
    FuseArgs fa;
    fa.sp_ids = sp_ids; fa.sp_rows = sp_rows; fa.n_sp = n_sp; fa.sp_src = sp_src; fa.sp_scale = sp_scale;
    fa.tg_ids = tg_ids; fa.tg_rows = tg_rows; fa.n_tg = n_tg; fa.wrow = wrow;
    fa.tile_meta = tile_meta;
    fa.emb1 = emb + H; fa.m1 = adam_m + H; fa.v1 = adam_v + H; fa.sh1w = nullptr;
    fa.lr_t = lr_t; fa.omb1 = 1.0f - beta1; fa.omb2 = 1.0f - beta2; fa.eps = eps;
    fa.extra1 = nullptr;
    int rc = tab16x3_launch(a, fa, (N + TI - 1) / TI, false, true, stream);
    if (rc) return rc;
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
