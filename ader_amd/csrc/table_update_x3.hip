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
#define HVB 32                     // gradient rows in flight per thread on the heavy path (k_tab32x3)
#define HVB1 16                    // ... of k_tab16x3 (168 registers: three workgroups per CU)
#define SPB 4                      // sparse-list entries per batch of the optimiser phase (loads of a batch are independent)
#ifndef AV
#define AV 6                       // 16-byte vectors per thread and load round of the optimiser phase (x theta, m, v)
#endif

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(8))) F16B { f32x4_t v; };       // 16-byte vector at an 8-byte aligned address

__device__ int x3_cu_arrivals[4096];
#ifdef T3_STAMP     // diagnostic build only (tools/build_variant.sh ... -DT3_STAMP): per-segment clocks of wave 0 of every workgroup
__device__ unsigned long long t3_dbg[12 * 1024];
#define STAMP(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                    __builtin_amdgcn_sched_barrier(0); seg[k_] += t_ - tprev; tprev = t_; }
extern "C" int ader_dbg_read(void* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(t3_dbg), (size_t)n * 8); }
#else
#define STAMP(k_)
#endif

// One chunk image (22 KiB, contiguous in memory) -> LDS buffer, as 22 LDS-DMA pieces of 1 KiB.
// The DMA is issued from inline asm ON PURPOSE: hipcc counts a __builtin_amdgcn_global_load_lds as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of the next ds_read_b64_tr_b16 it cannot tell apart from the destination -- in the middle of the
// chunk, which drained the prefetch half a chunk after it was issued.  Hidden from the compiler, the pieces are waited for by the
// explicit vmcnt(0) at the head of the next chunk only (cdna_hip_programming.md 5.7: M0 saved, written and restored in ONE statement).
// (scalar base + one 32-bit per-lane offset: no vector instruction per piece)
__device__ __forceinline__ void x3_glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// wave w moves pieces [6 w, 6 w + 6) of the 22 (wave 3: four): one scalar branch per chunk
__device__ __forceinline__ void x3_dma_chunk(const char* __restrict__ img, int c, bf16* buf, int wave_u, int lane) {
    const char* src0 = img + (size_t)c * X3_IMG_B + 6144 * wave_u;          // scalar
    const unsigned dst0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)buf) + 6144 * wave_u;
    const unsigned vo = 16 * lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) x3_glds16(src0 + 1024 * i, vo, dst0 + 1024 * i);
    if (wave_u < 3) {
        x3_glds16(src0 + 4096, vo, dst0 + 4096);
        x3_glds16(src0 + 5120, vo, dst0 + 5120);
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
#ifdef T3_STAMP
    unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#endif
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
    // ---- the first rep chunk is requested before anything else; the per-row constants and the tile's list record go to LDS
    const char* img = (const char*)a.rep_img;
    const int nch = a.Bp / X3_CH;
    if (!(a.ko & 2)) x3_dma_chunk(img, 0, R_l, wave, lane);
    // ---- operand fragments straight from memory: lane (item c16 of this wave's 16, k-group g) holds E[item][32 ks + 8 g + 0..7] as
    // hi + lo.  Two 16-byte loads per k-step at 8-byte aligned addresses; rows beyond the table's last one read zeros (range check);
    // channels >= H of the last k-step read what follows the row -- finite parameters that only ever meet the zero K-padding of rep.
    // (The round-3a kernel passed the tile through LDS: load, store, barrier, cut, barrier, and only then the first rep chunk --
    // three dependent latencies at the head of every tile.)
    bf16x8 e_hi[5], e_lo[5];
    {
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, (unsigned)n_av * 4u, 0x00020000);
        const int vt = 4 * ((wave * 16 + c16) * H + 8 * g);
        f32x4_t x0[5], x1[5];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            x0[ks] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rt, vt, 128 * ks, 0));
            x1[ks] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rt, vt, 128 * ks + 16, 0));
        }
        // (Measured and dropped: touching the theta rows of tile + 768 here -- one dword per 64 bytes, results unused, so that the next
        //  workgroup of the slot finds them on the chip -- made the kernel 3 % slower; so did prefetching theta / m / v of that tile by
        //  LDS-DMA from a fourth "prefetch" wave during the GEMM phase (+12 %): L2-hit rep pieces queue behind the HBM misses.)
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
        for (int ks = 0; ks < 5; ++ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (channels >= H: zero, so that the cut equals the one of a zero-padded row whatever follows the row in memory)
                const bool in0 = 32 * ks + 8 * g + j < H, in1 = 32 * ks + 8 * g + 4 + j < H;
                const float v0 = in0 ? x0[ks][j] : 0.0f, v1 = in1 ? x1[ks][j] : 0.0f;
                const bf16 h0 = (bf16)v0, h1 = (bf16)v1;
                e_hi[ks][j] = h0; e_hi[ks][4 + j] = h1;
                e_lo[ks][j] = (bf16)(v0 - (float)h0); e_lo[ks][4 + j] = (bf16)(v1 - (float)h1);
            }
        }
    }
    STAMP(0)
    __syncthreads();                                            // off_l / meta_l are in LDS
    STAMP(1)
    // the first input-embedding gradient rows of the tile (thread c holds column c), requested now, used after the GEMM
    float spv[SPV];
    // (every sparse-row product / sum below is kept as two rounded operations -- "#pragma clang fp contract(off)" -- so that the x3
    //  update kernels, which share this arithmetic, agree bit for bit whatever hipcc would fuse in each of them)
#pragma unroll
    for (int i = 0; i < SPV; ++i) {     // unconditional loads (row 0, column 0 when there is no entry): a load under a branch is waited
        const bool on = tid < H && meta_l[0] + i < meta_l[1];        // for at the end of its branch -- three round trips in a row
        { _Pragma("clang fp contract(off)") spv[i] = f.sp_src[on ? (size_t)meta_l[3 + 2 * i] * H + tid : 0] * (on ? f.sp_scale : 0.0f); }
    }
    STAMP(2)
    STAMP(3)
    f32x4v dE[10];
#pragma unroll
    for (int cb = 0; cb < 10; ++cb) dE[cb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const int q4 = c16 >> 2, p4 = c16 & 3;
    // per-lane byte offsets into a chunk image: row read of (row c16, k-group g); transposed read of (row 4g + q4, 4 channels p4)
    const int a_off = 1152 * (g >> 1) + 512 * (g & 1) + 16 * c16;
    const int t_off = 1152 * (p4 >> 1) + 16 * (4 * g + q4) + 8 * (p4 & 1);
    // (s_setprio 2 around this loop -- matrix phases ahead of the other workgroups' vector phases -- measured 2-3 % SLOWER: it starves
    //  the phases that request the next tile's memory)
    for (int c = 0; c < ((a.ko & 2) ? 0 : nch); ++c) {
        // this wave's pieces of chunk c have landed and its LDS reads of chunk c-1 are done; after the barrier that holds for every
        // wave, so chunk c can be read and the other buffer (chunk c-1's) can be refilled
        STAMP(4)
        if (!(a.ko & 8)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(5)
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
        // (the next chunk's DMA is issued behind the first operand reads: its scalar issue sequence runs under their latency)
        if (c + 1 < nch && !(a.ko & 4)) x3_dma_chunk(img, c + 1, R_l + ((c + 1) & 1) * X3_BUF, wave, lane);
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
        STAMP(6)
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
        STAMP(7)
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
    const float* __restrict__ gx = EXTRA ? f.extra1 + (size_t)tile0 * H : nullptr;
    // theta / m / v of the tile through buffer descriptors (base and size in scalar registers, one 32-bit per-lane offset, the
    // vector index as the scalar offset): vector u of thread t = floats 4 t + 1024 u of the block; floats >= n_el (the table's last,
    // partial tile; vectors 9.375.. of a full one) are range-checked away by the hardware, loads AND stores -- no per-vector branch.
    // ALL loads of the tile are in flight at once (m, v requested before the dE staging and the sparse terms, theta right behind
    // the staging, when the accumulators have left their registers): the phase waits out ONE memory latency instead of one per
    // round (stamps: 4-5 us each under load).
    const unsigned nbytes = (unsigned)n_el * 4u;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)gp, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)gm, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)gv, 0, nbytes, 0x00020000);
    const int vo = 16 * tid;
    f32x4_t P[NVEC], M[NVEC], V[NVEC];
#define LOAD_MV()                                                                                          \
    _Pragma("unroll") for (int u = 0; u < NVEC; ++u) {                                                     \
        M[u] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rm, vo, 4096 * u, 0));     \
        V[u] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 4096 * u, 0));     \
    }
#define LOAD_P()                                                                                           \
    _Pragma("unroll") for (int u = 0; u < NVEC; ++u)                                                       \
        P[u] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rp, vo, 4096 * u, 0));
    // A bucket that holds a hot item (Zipf ids: hundreds of entries) takes the HEAVY path below: its (id, row) lists are fetched
    // cooperatively, 256 entries per round trip, and the gradient rows HVB at a time -- with the optimiser loads requested AFTER
    // the sparse terms, so that the registers are free for the deeper batches (a workgroup-uniform choice; rare tiles).
    const bool heavy = (meta_l[1] - meta_l[0] > HEAVY_N) || (meta_l[TM_LIST + 1] - meta_l[TM_LIST] > HEAVY_N);
    STAMP(4)
    if (!heavy) { LOAD_MV(); }
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
    STAMP(8)
    {
    #pragma clang fp contract(off)
        // sparse terms of the tile: item ids [tile0+1, tile0+65).  Thread c owns column c of every row.
        const int id_lo = tile0 + 1, id_hi = min(tile0 + TI, N) + 1;
        if (heavy) {
            int* hv_l = (int*)(smem_raw + TI * HP * sizeof(float));     // [2][256] (id, row) of the current chunk, behind F_l
            // entries in list order (the same order, hence the same rounding, as the light path)
#define HEAVY_LIST(K0_, K1_, IDS_, ROWS_, VAL_, OP_)                                                       \
            for (int base_ = (K0_); base_ < (K1_); base_ += 256) {                                         \
                _Pragma("clang fp contract(off)")                                                            \
                const int n_ = min(256, (K1_) - base_);                                                    \
                if (tid < n_) { hv_l[tid] = (IDS_)[base_ + tid]; hv_l[256 + tid] = (ROWS_)[base_ + tid]; } \
                __syncthreads();                                                                           \
                if (tid < H && id_lo < id_hi) {                                                                \
                    /* the list is in (id, position) order: a table row's entries are consecutive -- its sum runs in a register, */ \
                    /* F + r1 + r2 + ... in position order exactly as one LDS update per entry gave it, without the chain of dependent */ \
                    /* LDS read-modify-writes (a hot item of the shipped data has hundreds of entries: +50 us on that tile's workgroup) */ \
                    int cur_ = -1;                                                                             \
                    float acc_ = 0.0f;                                                                         \
                    for (int e0_ = 0; e0_ < n_; e0_ += HVB1) {                                                  \
                        float val[HVB1];                                                                        \
                        _Pragma("unroll") for (int u = 0; u < HVB1; ++u) {                                      \
                            const int rw = (e0_ + u < n_) ? hv_l[256 + e0_ + u] : 0;                           \
                            val[u] = (VAL_);                                                                   \
                        }                                                                                      \
                        _Pragma("unroll") for (int u = 0; u < HVB1; ++u) {                                      \
                            const int id_ = (e0_ + u < n_) ? hv_l[e0_ + u] : 0x7fffffff;                       \
                            if (id_ < id_hi) {                                                                 \
                                if (id_ != cur_) {                                                             \
                                    if (cur_ >= 0) F_l[(cur_ - id_lo) * H + tid] = acc_;                       \
                                    cur_ = id_;                                                                \
                                    acc_ = F_l[(id_ - id_lo) * H + tid];                                       \
                                }                                                                              \
                                acc_ OP_ val[u];                                                               \
                            }                                                                                  \
                        }                                                                                      \
                    }                                                                                          \
                    if (cur_ >= 0) F_l[(cur_ - id_lo) * H + tid] = acc_;                                       \
                }                                                                                              \
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
    __builtin_amdgcn_sched_barrier(0);     // (theta after the sparse terms: their batches need the registers)
    if (heavy) { LOAD_MV(); }
    LOAD_P();
    lds_only_barrier();
    STAMP(9)
    // TF ApplyAdam (ADER.py:96): m += (g-m)(1-b1); v += (g*g-v)(1-b2); theta -= lr_t*m/(sqrt(v)+eps).  The square root and the
    // division use the hardware's v_sqrt_f32 / v_rcp_f32 (<= 1 ulp each; the update differs from the correctly rounded one by
    // < 4e-7 of ITSELF): the IEEE sequences cost ~27 vector instructions per element -- 1,300 of the 3,900 a wave issues per
    // tile, and it is vector ISSUE (matrix + vector instructions of three waves on one SIMD) that bounds this kernel (stamps and
    // counters: DESIGN.md section 6).  ADER_EXACT_DIV restores the correctly rounded forms.
#ifdef ADER_EXACT_DIV
#define ADAM1(p_, m_, v_, g_)                                                                              \
    { m_ += ((g_) - m_) * f.omb1; v_ += ((g_) * (g_) - v_) * f.omb2; p_ -= (m_ * f.lr_t) / (sqrtf(v_) + f.eps); }
#else
#define ADAM1(p_, m_, v_, g_)                                                                              \
    { m_ += ((g_) - m_) * f.omb1; v_ += ((g_) * (g_) - v_) * f.omb2;                                       \
      p_ -= (m_ * f.lr_t) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v_) + f.eps); }
#endif
#pragma unroll
    for (int u = 0; u < NVEC; ++u) {
        const int e = 4 * tid + 1024 * u;
        if (e < TI * H) {                                  // (the staging tile ends there; vector 9 exists for 96 threads)
            f32x4_t g4 = *(const f32x4_t*)(F_l + e);
            if (EXTRA) {
                const f32x4_t x4 = (e + 3 < n_el) ? ((const F16B*)(gx + e))->v : (f32x4_t){0.f, 0.f, 0.f, 0.f};
                g4 += x4;
            }
            f32x4_t p = P[u], m = M[u], v = V[u];
            ADAM1(p[0], m[0], v[0], g4[0]); ADAM1(p[1], m[1], v[1], g4[1]);
            ADAM1(p[2], m[2], v[2], g4[2]); ADAM1(p[3], m[3], v[3], g4[3]);
            // theta/m/v of this tile are not touched again this step: keep them out of the caches (aux 2 = nt)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, p), rp, vo, 4096 * u, 2);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, m), rm, vo, 4096 * u, 2);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rv, vo, 4096 * u, 2);
        }
    }
#undef ADAM1
#ifdef T3_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(10)
    if (tid == 0 && blockIdx.x % 16 == 5 && blockIdx.x / 16 < 1024) for (int k_ = 0; k_ < 12; ++k_) t3_dbg[(blockIdx.x / 16) * 12 + k_] = seg[k_];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_tab32x3: the same update with 128 table rows per workgroup -- each wave owns 16 rows of BOTH 64-row tiles of a pair -- and two
// workgroups per CU (<= 256 registers).  Why: every workgroup streams the WHOLE batch (512 rows x 160 channels, hi + lo = 352 KB)
// through its LDS once per tile; at 64 rows per workgroup that is 5.5 GB of L2 -> LDS traffic per 10^6 rows and ~1,000 clocks of
// LDS-DMA issue per 32-row chunk and wave (stamps: 11-18 % of a tile, wherever in the chunk the issue is placed).  With a pair of
// tiles per pass the DMA bytes, the DMA issue, the chunk barriers and the LDS operand reads (the rep fragments of a chunk feed the
// MFMAs of both tiles) are halved per table row.  The optimiser phase is k_tab16x3's, run once per tile of the pair.
template <bool EXTRA, bool KD>
__global__ __launch_bounds__(256, 2) void k_tab32x3(TabArgs a, FuseArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R_l = (bf16*)smem_raw;                                // [2 buffers][chunk image]; later the dE tile of one half
    float* off_l = (float*)(smem_raw + 2 * X3_IMG_B);           // [Bp]
    int* meta_l = (int*)(off_l + a.Bp);                         // the pair's list records [2 tiles][2][TM_LIST]
    float* toff_l = (float*)(meta_l + 4 * TM_LIST);             // KD: [Bp - kd_row0]
    int* trow_l = (int*)(toff_l + (a.Bp - a.kd_row0));          // KD: [Bp - kd_row0]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int H = a.H, N = a.N;
    const int tileA = 2 * blockIdx.x + a.tile_off;              // tiles tileA, tileA + 1 (the second may lie beyond the launch)
    const char* img = (const char*)a.rep_img;
    const int nch = a.Bp / X3_CH;
#ifdef T3_STAMP
    unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#endif
    x3_dma_chunk(img, 0, R_l, wave, lane);
    bf16x8 e_hi[2][5], e_lo[2][5];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tile0 = (tileA + h) * TI;
        const int rows_avail = (tileA + h < a.tile_end) ? min(TI, a.vrows - tile0) : 0;
        const int n_av = rows_avail > 0 ? rows_avail * H : 0;
        const size_t tqe = (a.ko & 256) ? (size_t)((tileA + h) & 63) * TI : (size_t)tile0;   // ko 256: operand rows served by L2 (timing only)
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)(a.emb1 + tqe * H), 0, (unsigned)n_av * 4u, 0x00020000);
        const int vt = 4 * ((wave * 16 + c16) * H + 8 * g);
        f32x4_t x0[5], x1[5];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            x0[ks] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rt, vt, 128 * ks, 0));
            x1[ks] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rt, vt, 128 * ks + 16, 0));
        }
        if (h == 0) {
            for (int i = tid; i < a.Bp; i += 256) off_l[i] = a.off[i];
            if (KD) {
                for (int i = tid; i < a.Bp - a.kd_row0; i += 256) {
                    const int b = a.kd_row0 + i, tr = a.trow[b];
                    const float w = f.wrow[b];
                    toff_l[i] = (tr >= 0 && w > 0.0f) ? log2f(w) - a.tlse2[b] : -INFINITY;
                    trow_l[i] = tr < 0 ? 0 : tr;
                }
            }
            if (tid < 4 * TM_LIST) {
                const int t_ = tileA + tid / (2 * TM_LIST);
                meta_l[tid] = (t_ < a.tile_end) ? f.tile_meta[(size_t)tileA * (2 * TM_LIST) + tid] : 0;
            }
        }
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in0 = 32 * ks + 8 * g + j < H, in1 = 32 * ks + 8 * g + 4 + j < H;
                const float v0 = in0 ? x0[ks][j] : 0.0f, v1 = in1 ? x1[ks][j] : 0.0f;
                const bf16 h0 = (bf16)v0, h1 = (bf16)v1;
                e_hi[h][ks][j] = h0; e_hi[h][ks][4 + j] = h1;
                e_lo[h][ks][j] = (bf16)(v0 - (float)h0); e_lo[h][ks][4 + j] = (bf16)(v1 - (float)h1);
            }
        }
    }
    STAMP(0)
    __syncthreads();                                            // off_l / meta_l are in LDS
    STAMP(1)
    float spv[2][SPV];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int* ms = meta_l + h * 2 * TM_LIST;
#pragma unroll
        for (int i = 0; i < SPV; ++i) {
            const bool on = tid < H && ms[0] + i < ms[1];
            { _Pragma("clang fp contract(off)") spv[h][i] = f.sp_src[on ? (size_t)ms[3 + 2 * i] * H + tid : 0] * (on ? f.sp_scale : 0.0f); }
        }
    }
    f32x4v dE[2][10];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int cb = 0; cb < 10; ++cb) dE[h][cb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    const int q4 = c16 >> 2, p4 = c16 & 3;
    const int a_off = 1152 * (g >> 1) + 512 * (g & 1) + 16 * c16;
    const int t_off = 1152 * (p4 >> 1) + 16 * (4 * g + q4) + 8 * (p4 & 1);
    const int itA = tileA * TI + wave * 16;                     // this wave's first item of half 0 (half 1: + TI)
    STAMP(2)
    for (int c = 0; c < nch; ++c) {
        STAMP(4)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(5)
        const char* Bh = (const char*)(R_l + (c & 1) * X3_BUF);
        const int b0 = c * X3_CH;
        float tv[KD ? 16 : 1];
        const bool kdc = KD && b0 >= a.kd_row0;
        if (kdc) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int it = itA + h * TI + c16;
                const bool okt = it < a.Np;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    tv[KD ? 8 * h + j : 0] = a.teacher[(size_t)trow_l[b0 - a.kd_row0 + 16 * (j >> 2) + 4 * g + (j & 3)] * a.ldt + (okt ? it : 0)];
            }
        }
#define X3_LOADA(set_, ks_)                                                                               \
        { const char* ap_ = Bh + a_off + X3_QUAD * (ks_);                                                 \
          set_[0] = *(const bf16x8*)ap_; set_[1] = *(const bf16x8*)(ap_ + X3_PLANE_B);                    \
          set_[2] = *(const bf16x8*)(ap_ + 256); set_[3] = *(const bf16x8*)(ap_ + X3_PLANE_B + 256); }
#define X3_LOADT(set_, cb_)                                                                               \
        { const bf16* tp_ = (const bf16*)(Bh + t_off + X3_QUAD * ((cb_) >> 1) + 512 * ((cb_) & 1));         \
          set_[0] = tr_read(tp_); set_[1] = tr_read(tp_ + 128);                                           \
          set_[2] = tr_read(tp_ + X3_PLANE_B / 2); set_[3] = tr_read(tp_ + X3_PLANE_B / 2 + 128); }
        f32x4v S[2][2];         // [half][row block]
#pragma unroll
        for (int h = 0; h < 2; ++h) { S[h][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; S[h][1] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
        bf16x8 fa[2][4];
        X3_LOADA(fa[0], 0);
        X3_LOADA(fa[1], 1);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < nch) x3_dma_chunk(img, c + 1, R_l + ((c + 1) & 1) * X3_BUF, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            bf16x8* A_ = fa[ks & 1];                             // {ah0, al0, ah1, al1}
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                S[h][0] = mfma16_bf16(A_[1], e_hi[h][ks], S[h][0]);
                S[h][1] = mfma16_bf16(A_[3], e_hi[h][ks], S[h][1]);
                S[h][0] = mfma16_bf16(A_[0], e_lo[h][ks], S[h][0]);
                S[h][1] = mfma16_bf16(A_[2], e_lo[h][ks], S[h][1]);
                S[h][0] = mfma16_bf16(A_[0], e_hi[h][ks], S[h][0]);
                S[h][1] = mfma16_bf16(A_[2], e_hi[h][ks], S[h][1]);
            }
            if (ks + 2 < 5) X3_LOADA(fa[ks & 1], ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(6)
        bf16x4 ft[3][4];
        X3_LOADT(ft[0], 0);
        X3_LOADT(ft[1], 1);
        X3_LOADT(ft[2], 2);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 ph_[2], pl_[2];  // k order of a fragment: rows 4g..4g+3 of row block 0, then of row block 1
        {
            const float4 o0 = *(const float4*)(off_l + b0 + 4 * g);
            const float4 o1 = *(const float4*)(off_l + b0 + 16 + 4 * g);
            const float o0a[4] = {o0.x, o0.y, o0.z, o0.w}, o1a[4] = {o1.x, o1.y, o1.z, o1.w};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    S[h][0][j] = __builtin_amdgcn_exp2f(fmaf(S[h][0][j], LOG2E, o0a[j]));
                    S[h][1][j] = __builtin_amdgcn_exp2f(fmaf(S[h][1][j], LOG2E, o1a[j]));
                }
                if (kdc) {
                    if (itA + h * TI + c16 < a.Np) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            S[h][0][j] -= __builtin_amdgcn_exp2f(fmaf(tv[KD ? 8 * h + j : 0], LOG2E, toff_l[b0 - a.kd_row0 + 4 * g + j]));
                            S[h][1][j] -= __builtin_amdgcn_exp2f(fmaf(tv[KD ? 8 * h + 4 + j : 0], LOG2E, toff_l[b0 - a.kd_row0 + 16 + 4 * g + j]));
                        }
                    } else {
                        S[h][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; S[h][1] = (f32x4v){0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16 h0 = (bf16)S[h][0][j], h1 = (bf16)S[h][1][j];
                    ph_[h][j] = h0; ph_[h][4 + j] = h1;
                    pl_[h][j] = (bf16)(S[h][0][j] - (float)h0); pl_[h][4 + j] = (bf16)(S[h][1][j] - (float)h1);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(7)
#pragma unroll
        for (int cb = 0; cb < 10; ++cb) {
            bf16x4* T_ = ft[cb % 3];
            bf16x8 bh, bl;
#pragma unroll
            for (int j = 0; j < 4; ++j) { bh[j] = T_[0][j]; bh[4 + j] = T_[1][j]; bl[j] = T_[2][j]; bl[4 + j] = T_[3][j]; }
            dE[0][cb] = mfma16_bf16(pl_[0], bh, dE[0][cb]);
            dE[1][cb] = mfma16_bf16(pl_[1], bh, dE[1][cb]);
            dE[0][cb] = mfma16_bf16(ph_[0], bl, dE[0][cb]);
            dE[1][cb] = mfma16_bf16(ph_[1], bl, dE[1][cb]);
            dE[0][cb] = mfma16_bf16(ph_[0], bh, dE[0][cb]);
            dE[1][cb] = mfma16_bf16(ph_[1], bh, dE[1][cb]);
            if (cb + 3 < 10) X3_LOADT(ft[cb % 3], cb + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef X3_LOADA
#undef X3_LOADT
    // ---- optimiser phase, once per tile of the pair (k_tab16x3's: comments there)
    float* F_l = (float*)smem_raw;
    const int vo = 16 * tid;
    STAMP(4)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tile = tileA + h;
        const int tile0 = tile * TI;
        const int* ms = meta_l + h * 2 * TM_LIST;
        const int* mg = ms + TM_LIST;
        const int rows_valid = (tile < a.tile_end) ? min(TI, N - tile0) : 0;
        const int n_el = rows_valid > 0 ? rows_valid * H : 0;
        float* __restrict__ gp = f.emb1 + (size_t)tile0 * H;
        float* __restrict__ gm = f.m1 + (size_t)tile0 * H;
        float* __restrict__ gv = f.v1 + (size_t)tile0 * H;
        const float* __restrict__ gx = EXTRA ? f.extra1 + (size_t)tile0 * H : nullptr;
        const unsigned nbytes = (unsigned)n_el * 4u;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)gp, 0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)gm, 0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)gv, 0, nbytes, 0x00020000);
        f32x4_t P[NVEC], M[NVEC], V[NVEC];
        const bool heavy = (ms[1] - ms[0] > HEAVY_N) || (mg[1] - mg[0] > HEAVY_N);
        // (measured and dropped: theta requested here too, with m and v -- the registers are there at two workgroups per CU -- made the
        //  step 0.04 ms SLOWER: 30 loads issued ahead of the staging stores delay them, and the Adam section did not get shorter)
        if (!heavy) { LOAD_MV(); }
        lds_only_barrier();             // every wave is done with the last rep chunk / with the other half's staging tile
#pragma unroll
        for (int cb = 0; cb < 10; ++cb) {
            const int hc = 32 * (cb >> 1) + 8 * (cb & 1) + 16 * (c16 >> 3) + (c16 & 7);
            if (hc < H) {
#pragma unroll
                for (int j = 0; j < 4; ++j) F_l[(wave * 16 + 4 * g + j) * H + hc] = dE[h][cb][j];
            }
        }
        lds_only_barrier();
        STAMP(8)
        {
        #pragma clang fp contract(off)
            const int id_lo = tile0 + 1, id_hi = (tile < a.tile_end ? min(tile0 + TI, N) : tile0) + 1;
            if (heavy) {
                int* hv_l = (int*)(smem_raw + TI * HP * sizeof(float));
#define HEAVY_LIST(K0_, K1_, IDS_, ROWS_, VAL_, OP_)                                                       \
                for (int base_ = (K0_); base_ < (K1_); base_ += 256) {                                     \
                    _Pragma("clang fp contract(off)")                                                        \
                    const int n_ = min(256, (K1_) - base_);                                                \
                    if (tid < n_) { hv_l[tid] = (IDS_)[base_ + tid]; hv_l[256 + tid] = (ROWS_)[base_ + tid]; } \
                    __syncthreads();                                                                       \
                    if (tid < H && id_lo < id_hi) {                                                                \
                        /* the list is in (id, position) order: a table row's entries are consecutive -- its sum runs in a register, */ \
                        /* F + r1 + r2 + ... in position order exactly as one LDS update per entry gave it, without the chain of dependent */ \
                        /* LDS read-modify-writes (a hot item of the shipped data has hundreds of entries: +50 us on that tile's workgroup) */ \
                        int cur_ = -1;                                                                             \
                        float acc_ = 0.0f;                                                                         \
                        for (int e0_ = 0; e0_ < n_; e0_ += HVB) {                                                  \
                            float val[HVB];                                                                        \
                            _Pragma("unroll") for (int u = 0; u < HVB; ++u) {                                      \
                                const int rw = (e0_ + u < n_) ? hv_l[256 + e0_ + u] : 0;                           \
                                val[u] = (VAL_);                                                                   \
                            }                                                                                      \
                            _Pragma("unroll") for (int u = 0; u < HVB; ++u) {                                      \
                                const int id_ = (e0_ + u < n_) ? hv_l[e0_ + u] : 0x7fffffff;                       \
                                if (id_ < id_hi) {                                                                 \
                                    if (id_ != cur_) {                                                             \
                                        if (cur_ >= 0) F_l[(cur_ - id_lo) * H + tid] = acc_;                       \
                                        cur_ = id_;                                                                \
                                        acc_ = F_l[(id_ - id_lo) * H + tid];                                       \
                                    }                                                                              \
                                    acc_ OP_ val[u];                                                               \
                                }                                                                                  \
                            }                                                                                      \
                        }                                                                                          \
                        if (cur_ >= 0) F_l[(cur_ - id_lo) * H + tid] = acc_;                                       \
                    }                                                                                              \
                    __syncthreads();                                                                       \
                }
                HEAVY_LIST(ms[0], ms[1], f.sp_ids, f.sp_rows, f.sp_src[(size_t)rw * H + tid] * f.sp_scale, +=)
                HEAVY_LIST(mg[0], mg[1], f.tg_ids, f.tg_rows,
                           f.wrow[rw] * ((float)a.rep_hi[(size_t)rw * LDR + tid] + (float)a.rep_lo[(size_t)rw * LDR + tid]), -=)
#undef HEAVY_LIST
            } else if (tid < H && id_lo < id_hi) {
                const int k0s = ms[0], k1s = ms[1];
#pragma unroll
                for (int i = 0; i < SPV; ++i) {
                    if (k0s + i < k1s) {
                        const int id = ms[2 + 2 * i];
                        if (id < id_hi) F_l[(id - id_lo) * H + tid] += spv[h][i];
                    }
                }
                for (int k = k0s + SPV, i = SPV; k < k1s; k += SPB, i += SPB) {
                    int idv[SPB], rw[SPB];
                    float val[SPB];
#pragma unroll
                    for (int u = 0; u < SPB; ++u) {
                        const int ic = (i + u) < 8 ? (i + u) : 7;
                        const int id_c = ms[2 + 2 * ic], row_c = ms[3 + 2 * ic];
                        const bool in = k + u < k1s;
                        int id_g = 0, row_g = 0;
                        if (i + SPB > 8) {
                            const int ke = in ? k + u : k0s;
                            id_g = f.sp_ids[ke]; row_g = f.sp_rows[ke];
                        }
                        idv[u] = !in ? 0x7fffffff : ((i + u < 8) ? id_c : id_g);
                        rw[u] = !in ? 0 : ((i + u < 8) ? row_c : row_g);
                    }
#pragma unroll
                    for (int u = 0; u < SPB; ++u)
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
                        const float rv_ = (float)a.rep_hi[(size_t)bw[u] * LDR + tid] + (float)a.rep_lo[(size_t)bw[u] * LDR + tid];
                        val[u] = rv_ * f.wrow[bw[u]] * ((idv[u] < id_hi) ? 1.0f : 0.0f);
                    }
#pragma unroll
                    for (int u = 0; u < SPB; ++u)
                        if (idv[u] < id_hi) F_l[(idv[u] - id_lo) * H + tid] -= val[u];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (heavy) { LOAD_MV(); }
        LOAD_P();
        lds_only_barrier();
        STAMP(9)
#define ADAM1(p_, m_, v_, g_)                                                                              \
        { m_ += ((g_) - m_) * f.omb1; v_ += ((g_) * (g_) - v_) * f.omb2;                                   \
          p_ -= (m_ * f.lr_t) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v_) + f.eps); }
#pragma unroll
        for (int u = 0; u < NVEC; ++u) {
            const int e = 4 * tid + 1024 * u;
            if (e < TI * H) {
                f32x4_t g4 = *(const f32x4_t*)(F_l + e);
                if (EXTRA) {
                    const f32x4_t x4 = (e + 3 < n_el) ? ((const F16B*)(gx + e))->v : (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    g4 += x4;
                }
                f32x4_t p = P[u], m = M[u], v = V[u];
                ADAM1(p[0], m[0], v[0], g4[0]); ADAM1(p[1], m[1], v[1], g4[1]);
                ADAM1(p[2], m[2], v[2], g4[2]); ADAM1(p[3], m[3], v[3], g4[3]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, p), rp, vo, 4096 * u, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, m), rm, vo, 4096 * u, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rv, vo, 4096 * u, 2);
            }
        }
#undef ADAM1
#ifdef T3_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(10)
#endif
    }
#ifdef T3_STAMP
    if (tid == 0 && blockIdx.x % 8 == 5 && blockIdx.x / 8 < 1024) for (int k_ = 0; k_ < 12; ++k_) t3_dbg[(blockIdx.x / 8) * 12 + k_] = seg[k_];
#endif
}

// ============================================================================================= launch (C ABI: table_update.hip)
// Timing-only knobs of the diagnostic builds (tools/build_variant.sh ... -DADER_DIAG): knock-outs, staggers, an LDS pad and the
// one-tile kernel.  A production build reads NO environment variable here: a stray one must not be able to corrupt a training run.
static size_t tab16x3_lds(int Bp, int Bk) {
    int pad = 0;
#ifdef ADER_DIAG
    { static int pad_ = -1; if (pad_ < 0) { const char* e = getenv("ADER_X3_LDSPAD"); pad_ = e ? atoi(e) : 0; } pad = pad_; }
#endif
    return (size_t)2 * X3_IMG_B + (size_t)Bp * sizeof(float) + 4 * TM_LIST * sizeof(int) + (size_t)Bk * 8 + pad;
}
static bool tab_pairs() {                 // (ADER_DIAG, ADER_X3_TILE=64: k_tab16x3 -- one 64-row tile per workgroup -- instead of k_tab32x3)
#ifdef ADER_DIAG
    static int v = -1;
    if (v < 0) { const char* e = getenv("ADER_X3_TILE"); v = (e && atoi(e) == 64) ? 0 : 1; }
    return v == 1;
#else
    return true;
#endif
}

static int g_x3_pair_min_tiles = 0;       // ader_x3_update_pair_min_tiles(): 0 = pairs always (measured: no difference on the shipped catalogs)

template <bool EXTRA, bool KD>
static int tab16x3_launch_t(const TabArgs& a, const FuseArgs& fa, int tiles, size_t lds, hipStream_t st) {
    static int lds_set_dev[ADER_MAX_DEV] = {};
    int& lds_set = lds_set_dev[ader_cur_dev()];
    if ((int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_tab16x3<EXTRA, KD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = (int)lds;
    }
    hipLaunchKernelGGL((k_tab16x3<EXTRA, KD>), dim3(tiles), dim3(256), lds, st, a, fa);
    return 0;
}
template <bool EXTRA, bool KD>
static int tab32x3_launch_t(const TabArgs& a, const FuseArgs& fa, int tiles, size_t lds, hipStream_t st) {
    static int lds_set_dev[ADER_MAX_DEV] = {};
    int& lds_set = lds_set_dev[ader_cur_dev()];
    if ((int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_tab32x3<EXTRA, KD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = (int)lds;
    }
    hipLaunchKernelGGL((k_tab32x3<EXTRA, KD>), dim3((tiles + 1) / 2), dim3(256), lds, st, a, fa);
    return 0;
}

static int tab16x3_launch(TabArgs a, const FuseArgs& fa, int tiles, bool extra, bool kd, void* stream) {
    a.ko = 0;
#ifdef ADER_DIAG
    { static int ko = -1; if (ko < 0) { const char* e = getenv("ADER_X3_KO"); ko = e ? atoi(e) : 0; const char* s_ = getenv("ADER_X3_STAGGER"); ko |= (s_ ? atoi(s_) : 0) << 16; } a.ko = ko; }
#endif
    if (a.Bp % X3_CH != 0 || (kd && a.kd_row0 % X3_CH != 0) || !a.rep_img || ((uintptr_t)a.rep_img & 15)) return -2;
    const size_t lds = tab16x3_lds(a.Bp, kd ? a.Bp - a.kd_row0 : 0);
    hipStream_t st = (hipStream_t)stream;
    a.tile_end = a.tile_off + tiles;
    // (A/B knob: below g_x3_pair_min_tiles tiles, one tile per workgroup.  Measured on the shipped catalogs, 400-700 tiles, distilled step:
    //  k_tab16x3 107.6 us against k_tab32x3 112.1 us, step 0.4594 against 0.4575 ms -- no difference, the default stays pairs)
    if (tab_pairs() && tiles > g_x3_pair_min_tiles && (a.tile_off & 1) == 0 && !(a.ko & 0xff)) {
        if (kd) return tab32x3_launch_t<false, true>(a, fa, tiles, lds, st);
        if (extra) return tab32x3_launch_t<true, false>(a, fa, tiles, lds, st);
        return tab32x3_launch_t<false, false>(a, fa, tiles, lds, st);
    }
    if (kd) return tab16x3_launch_t<false, true>(a, fa, tiles, lds, st);
    if (extra) return tab16x3_launch_t<true, false>(a, fa, tiles, lds, st);
    return tab16x3_launch_t<false, false>(a, fa, tiles, lds, st);
}

extern "C" {

// (The role-split producer / consumer form of this update -- k_tabp, rounds 4-5, opt-in -- was removed in round 5: it never won in the
//  step and the full-size bit-identity test caught it writing a few wrong vectors in one launch out of several on a
//  cold process: a kernel with a rare race has no place behind the ABI.)
// tiles (64 table rows each) above which the update takes a PAIR of tiles per workgroup (k_tab32x3) instead of one (k_tab16x3):
// tuning / A-B knob, bit-identical results either way; negative: query.  Returns the previous setting.
int ader_x3_update_pair_min_tiles(int tiles) {
    const int prev = g_x3_pair_min_tiles;
    if (tiles >= 0) g_x3_pair_min_tiles = tiles;
    return prev;
}

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

// ... and for a DISTILLED step (ADER.py:132-137): arguments as ader_tab_update_kd plus rep_img; the _range form restricts the update to
// the 128-item tiles [tile_begin, tile_begin + tile_count) (tile_count < 0: all) -- a rank's shard of a catalog-sharded table, with
// Bp / kd_row0 / off / wrow / trow / tlse2 describing the GLOBAL batch ([all train rows | all exemplar rows]).
int ader_tab_update_x3_kd_range(const void* rep_hi, const void* rep_lo, const void* rep_img, int item_num, int Bp, int kd_row0, int H,
                                int N, int Np, const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src,
                                float sp_scale, const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta,
                                const float* wrow, const float* teacher, long ldt, const int* trow, const float* tlse2, float* emb,
                                float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                                int tile_count, void* stream);
int ader_tab_update_x3_kd(const void* rep_hi, const void* rep_lo, const void* rep_img, int item_num, int Bp, int kd_row0, int H, int N,
                          int Np, const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src,
                          float sp_scale, const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta, const float* wrow,
                          const float* teacher, long ldt, const int* trow, const float* tlse2, float* emb, float* adam_m,
                          float* adam_v, float lr_t, float beta1, float beta2, float eps, void* stream) {
    return ader_tab_update_x3_kd_range(rep_hi, rep_lo, rep_img, item_num, Bp, kd_row0, H, N, Np, off, sp_ids, sp_rows, n_sp, sp_src, sp_scale,
                                       tg_ids, tg_rows, n_tg, tile_meta, wrow, teacher, ldt, trow, tlse2, emb, adam_m, adam_v, lr_t, beta1,
                                       beta2, eps, 0, -1, stream);
}
int ader_tab_update_x3_kd_range(const void* rep_hi, const void* rep_lo, const void* rep_img, int item_num, int Bp, int kd_row0, int H,
                                int N, int Np, const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src,
                                float sp_scale, const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta,
                                const float* wrow, const float* teacher, long ldt, const int* trow, const float* tlse2, float* emb,
                                float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                                int tile_count, void* stream) {
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
    const int all = (N + TI - 1) / TI;
    int tb = (tile_begin < 0 ? 0 : tile_begin) * 2;
    int te = tile_count < 0 ? all : tb + tile_count * 2;
    if (te > all) te = all;
    if (te <= tb) return 0;
    a.tile_off = tb;
    int rc = tab16x3_launch(a, fa, te - tb, false, true, stream);
    if (rc) return rc;
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
