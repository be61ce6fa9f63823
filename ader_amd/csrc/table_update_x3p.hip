// k_tabp: the float32-grade fused table update (table-gradient GEMM + sparse rows + dense TF-Adam, ADER.py:91-96) as a
// PIPELINE of specialised waves inside persistent workgroups.  Same arithmetic, operand images and lists as k_tab32x3
// (table_update_x3.hip); results are bit-identical to it.
//
// Why: k_tab32x3's workgroups alternate a matrix phase (65 % of their time: S = rep.E^T recomputed, dE += P^T.rep) and an
// optimiser phase (35 %: theta / m / v of 128 table rows in and out of HBM), two workgroups per CU (243 registers).  Neither
// pipe is full -- HBM 0.45 of peak, matrix pipe 54 % busy -- because each resource idles while both residents are in the other
// phase, and the registers leave no room for a third workgroup.  Here ONE 512-thread workgroup per CU (persistent, 256 registers)
// runs the two phases AT THE SAME TIME on different waves, one pair of 64-row tiles apart:
//   waves 0-3  GEMM    one per SIMD, each owns 16 rows of both tiles of the pair: S and dE on v_mfma_f32_16x16x32_bf16 exactly as in
//                      k_tab32x3 -- but they issue no memory instruction inside the chunk loop (the LDS-DMA issue cost them ~1,000 of
//                      a chunk's ~2,900 clocks there); at the end of a pair they request the NEXT pair's theta rows (operand cut),
//                      hand dE to the optimiser waves through LDS and go on.
//   waves 4-5  LOADER  stream the rep chunk images by LDS-DMA into a ring of THREE buffers, two chunks ahead of the GEMM waves.
//   waves 6-7  ADAM    the previous pair: sparse input-embedding / one-hot rows into the staged dE tile, then TF-Adam over the pair's
//                      theta / m / v in float4 rounds spread evenly over the chunk slots of the GEMM waves' current pair, each
//                      round's loads requested two slots ahead -- the HBM stream runs at a constant rate under the matrix work.
// All eight waves meet at one s_barrier per chunk (+ two per pair around the dE hand-off).  Used for large catalogs (every CU gets
// several pairs); small ones, distilled steps and the EXTRA form keep k_tab32x3.  gfx950 only.
#include "lbf_common.h"
#include "x3_image.h"
#include "../../include/ader_hip.h"

#define TI 64
#define X3_CH 32
#define TM_LIST 18
#define SPV 3
#define TP_NBUF 3                  // rep chunk buffers
#define TP_MR 4                    // Adam rounds (128 threads x float4) per chunk slot, at most
#define TP_AT 128                  // Adam threads

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tp_glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// loader wave dw (0 / 1) moves pieces [11 dw, 11 dw + 11) of a chunk image (22 pieces of 1 KiB)
__device__ __forceinline__ void tp_dma_chunk(const char* __restrict__ img, int chunk, bf16* buf, int dw, int lane) {
    const char* src0 = img + (size_t)chunk * X3_IMG_B + 11264 * dw;
    const unsigned dst0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)buf) + 11264 * dw;
    const unsigned vo = 16 * lane;
#pragma unroll
    for (int i = 0; i < 11; ++i) tp_glds16(src0 + 1024 * i, vo, dst0 + 1024 * i);
}

#ifdef ADER_EXACT_DIV
#define TP_ADAM1(p_, m_, v_, g_)                                                                           \
    { m_ += ((g_) - m_) * f.omb1; v_ += ((g_) * (g_) - v_) * f.omb2; p_ -= (m_ * f.lr_t) / (sqrtf(v_) + f.eps); }
#else
#define TP_ADAM1(p_, m_, v_, g_)                                                                           \
    { m_ += ((g_) - m_) * f.omb1; v_ += ((g_) * (g_) - v_) * f.omb2;                                       \
      p_ -= (m_ * f.lr_t) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v_) + f.eps); }
#endif

__global__ __launch_bounds__(512) void k_tabp(TabArgs a, FuseArgs f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R_l = (bf16*)smem_raw;                                            // [TP_NBUF][chunk image]
    float* F_l = (float*)(smem_raw + TP_NBUF * X3_IMG_B);                   // [2 tiles][64 * H] staged dE of the handed-off pair
    float* off_l = F_l + 2 * TI * 150;                                      // [Bp]
    int* meta_l = (int*)(off_l + a.Bp);                                     // [2][4 * TM_LIST] list records of two pairs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, N = a.N;
    const int nch = a.Bp / X3_CH;
    const int G = gridDim.x;
    const int npairs = (a.tile_end - a.tile_off + 1) / 2;
    const int n_it = (npairs - (int)blockIdx.x + G - 1) / G;                // pairs of this workgroup: blockIdx.x + G i
    const char* img = (const char*)a.rep_img;
    for (int i = tid; i < a.Bp; i += 512) off_l[i] = a.off[i];
    __syncthreads();

    if (wave < 4) {
        // =========================================================================================== GEMM waves
        const int c16 = lane & 15, g = lane >> 4;
        const int q4 = c16 >> 2, p4 = c16 & 3;
        const int a_off = 1152 * (g >> 1) + 512 * (g & 1) + 16 * c16;
        const int t_off = 1152 * (p4 >> 1) + 16 * (4 * g + q4) + 8 * (p4 & 1);
        bf16x8 e_hi[2][5], e_lo[2][5];
        f32x4_t x0[2][5], x1[2][5];
#define TP_LOAD_X(pair_)                                                                                   \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                    \
            const int tl_ = a.tile_off + 2 * (pair_) + h;                                                  \
            const int tile0_ = tl_ * TI;                                                                   \
            const int ra_ = (tl_ < a.tile_end) ? min(TI, a.vrows - tile0_) : 0;                            \
            const int nav_ = ra_ > 0 ? ra_ * H : 0;                                                        \
            const __amdgpu_buffer_rsrc_t rt_ = __builtin_amdgcn_make_buffer_rsrc((void*)(a.emb1 + (size_t)(nav_ ? tile0_ : 0) * H), 0, \
                                                                                 (unsigned)nav_ * 4u, 0x00020000);       \
            const int vt_ = 4 * ((wave * 16 + c16) * H + 8 * g);                                           \
            _Pragma("unroll") for (int ks = 0; ks < 5; ++ks) {                                             \
                x0[h][ks] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rt_, vt_, 128 * ks, 0));      \
                x1[h][ks] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rt_, vt_, 128 * ks + 16, 0)); \
            }                                                                                              \
        }
#define TP_CUT_X()                                                                                         \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 5; ++ks)                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
            const bool in0 = 32 * ks + 8 * g + j < H, in1 = 32 * ks + 8 * g + 4 + j < H;                   \
            const float v0 = in0 ? x0[h][ks][j] : 0.0f, v1 = in1 ? x1[h][ks][j] : 0.0f;                    \
            const bf16 h0 = (bf16)v0, h1 = (bf16)v1;                                                       \
            e_hi[h][ks][j] = h0; e_hi[h][ks][4 + j] = h1;                                                  \
            e_lo[h][ks][j] = (bf16)(v0 - (float)h0); e_lo[h][ks][4 + j] = (bf16)(v1 - (float)h1);          \
        }
        TP_LOAD_X((int)blockIdx.x);
        TP_CUT_X();
        f32x4v dE[2][10];
        int kb = 0;                                                         // ring buffer of the chunk about to be read
        for (int it = 0; it <= n_it; ++it) {
            const bool act = it < n_it;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int cb = 0; cb < 10; ++cb) dE[h][cb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < nch; ++c) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (act) {
                    const char* Bh = (const char*)(R_l + kb * X3_BUF);
                    const int b0 = c * X3_CH;
#define X3_LOADA(set_, ks_)                                                                               \
                    { const char* ap_ = Bh + a_off + X3_QUAD * (ks_);                                     \
                      set_[0] = *(const bf16x8*)ap_; set_[1] = *(const bf16x8*)(ap_ + X3_PLANE_B);        \
                      set_[2] = *(const bf16x8*)(ap_ + 256); set_[3] = *(const bf16x8*)(ap_ + X3_PLANE_B + 256); }
#define X3_LOADT(set_, cb_)                                                                               \
                    { const bf16* tp_ = (const bf16*)(Bh + t_off + X3_QUAD * ((cb_) >> 1) + 512 * ((cb_) & 1)); \
                      set_[0] = tr_read(tp_); set_[1] = tr_read(tp_ + 128);                               \
                      set_[2] = tr_read(tp_ + X3_PLANE_B / 2); set_[3] = tr_read(tp_ + X3_PLANE_B / 2 + 128); }
                    f32x4v S[2][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) { S[h][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; S[h][1] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
                    bf16x8 fa[2][4];
                    X3_LOADA(fa[0], 0);
                    X3_LOADA(fa[1], 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) {
                        bf16x8* A_ = fa[ks & 1];
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
                    bf16x4 ft[3][4];
                    X3_LOADT(ft[0], 0);
                    X3_LOADT(ft[1], 1);
                    X3_LOADT(ft[2], 2);
                    __builtin_amdgcn_sched_barrier(0);
                    bf16x8 ph_[2], pl_[2];
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
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bf16 h0 = (bf16)S[h][0][j], h1 = (bf16)S[h][1][j];
                                ph_[h][j] = h0; ph_[h][4 + j] = h1;
                                pl_[h][j] = (bf16)(S[h][0][j] - (float)h0); pl_[h][4 + j] = (bf16)(S[h][1][j] - (float)h1);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
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
#undef X3_LOADA
#undef X3_LOADT
                }
                kb = (kb == TP_NBUF - 1) ? 0 : kb + 1;
            }
            // the next pair's theta rows (operand cut), requested before the hand-off so that their latency runs under it
            // (unconditional: beyond the workgroup's last pair the descriptor has zero records -- a load under a branch would be
            //  waited for at the end of that branch)
            TP_LOAD_X((int)blockIdx.x + G * (it + 1));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // H1: the Adam waves are done with the staged tile
            if (act) {
                // element (row wave*16 + 4g + j, channel hc(cb)) of tile h; the lane part of the index is made opaque once per pair:
                // left visible, hipcc hoists all 80 store addresses out of the pair loop, spills them, and reloads each one behind
                // an s_waitcnt vmcnt(0) -- i.e. behind the next pair's theta loads
                int fb = (wave * 16 + 4 * g) * H + 16 * (c16 >> 3) + (c16 & 7);
                asm volatile("" : "+v"(fb));
                const int hl = 16 * (c16 >> 3) + (c16 & 7);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float* row = F_l + h * TI * H + j * H + fb;
#pragma unroll
                        for (int cb = 0; cb < 10; ++cb) {
                            const int hb = 32 * (cb >> 1) + 8 * (cb & 1);
                            if (hb + hl < H) row[hb] = dE[h][cb][j];
                        }
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // H2: dE of this pair is staged
            TP_CUT_X();
        }
#undef TP_LOAD_X
#undef TP_CUT_X
    } else if (wave < 6) {
        // =========================================================================================== loader waves
        const int dw = wave - 4;
        const int total = n_it * nch;                                       // chunks the GEMM waves will read
        if (total > 0) tp_dma_chunk(img, 0, R_l, dw, lane);
        if (total > 1) tp_dma_chunk(img, 1 % nch, R_l + X3_BUF, dw, lane);
        int k = 0, kb2 = 2 % TP_NBUF, ch2 = 2 % nch;                        // global chunk counter; buffer / image of chunk k + 2
        for (int it = 0; it <= n_it; ++it) {
            for (int c = 0; c < nch; ++c) {
                // chunk k has landed (the 11 pieces of chunk k + 1 may still be in flight)
                if (k + 1 < total) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                // behind barrier k every GEMM wave has finished reading chunk k - 1: its buffer takes chunk k + 2
                if (k + 2 < total) tp_dma_chunk(img, ch2, R_l + kb2 * X3_BUF, dw, lane);
                ++k;
                kb2 = (kb2 == TP_NBUF - 1) ? 0 : kb2 + 1;
                ch2 = (ch2 == nch - 1) ? 0 : ch2 + 1;
            }
            asm volatile("s_barrier" ::: "memory");                         // H1
            asm volatile("s_barrier" ::: "memory");                         // H2
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // =========================================================================================== Adam waves
        const int at = tid - 384;                                           // 0..127
        const int RT = (TI * H + 4 * TP_AT - 1) / (4 * TP_AT);              // float4 rounds of 128 threads per 64-row tile
        const int R = 2 * RT;                                               // ... per pair
        const int col0 = at, col1 = at + TP_AT;
        const bool has1 = col1 < H;
        f32x4_t P[2][TP_MR], M[2][TP_MR], V[2][TP_MR];
        float spv[2][SPV][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < SPV; ++i) { spv[h][i][0] = 0.0f; spv[h][i][1] = 0.0f; }
        int mreg = 0;
        const int vo = 16 * at;
        // descriptor of tile h of pair index pi (num_records 0: the pair does not exist / lies beyond the launch -> loads read 0,
        // stores are dropped by the hardware range check: no branch around any memory instruction)
#define TP_DESC(r_, base_, pi_, h_, ok_)                                                                   \
        { const int tl_ = a.tile_off + 2 * (pi_) + (h_);                                                   \
          const int tile0_ = tl_ * TI;                                                                     \
          const int rv_ = ((ok_) && tl_ < a.tile_end) ? min(TI, N - tile0_) : 0;                           \
          const unsigned nb_ = rv_ > 0 ? (unsigned)(rv_ * H) * 4u : 0u;                                     \
          r_ = __builtin_amdgcn_make_buffer_rsrc((void*)((base_) + (size_t)(nb_ ? tile0_ : 0) * H), 0, nb_, 0x00020000); }
        // rounds [cum(s - 1), cum(s)) of the pair go to slot s (1 <= s <= nch - 1; slot 0 applies the sparse rows)
        const int den = nch - 1;
#define TP_CUM(s_) (((s_) * R) / den)
        // issue the loads of (pair pi_, slot s_) into register set set_
#define TP_ISSUE(set_, pi_, s_, ok_)                                                                       \
        { const int r0_ = ((s_) >= 1) ? TP_CUM((s_) - 1) : 0;                                              \
          const int r1_ = ((s_) >= 1) ? TP_CUM(s_) : 0;                                                    \
          _Pragma("unroll") for (int j = 0; j < TP_MR; ++j) {                                              \
              const int r_ = r0_ + j;                                                                      \
              const bool on_ = (ok_) && r_ < r1_;                                                          \
              const int h_ = r_ >= RT ? 1 : 0;                                                             \
              const int u_ = r_ - h_ * RT;                                                                 \
              __amdgpu_buffer_rsrc_t rp_, rm_, rv2_;                                                       \
              TP_DESC(rp_, f.emb1, pi_, h_, on_); TP_DESC(rm_, f.m1, pi_, h_, on_); TP_DESC(rv2_, f.v1, pi_, h_, on_); \
              P[set_][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rp_, vo, 2048 * u_, 0));   \
              M[set_][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rm_, vo, 2048 * u_, 0));   \
              V[set_][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rv2_, vo, 2048 * u_, 0));  \
          } }
        // Adam on the rounds of (pair pi_, slot s_) held in register set set_
#define TP_ROUNDS(set_, pi_, s_, ok_)                                                                      \
        { const int r0_ = ((s_) >= 1) ? TP_CUM((s_) - 1) : 0;                                              \
          const int r1_ = ((s_) >= 1) ? TP_CUM(s_) : 0;                                                    \
          _Pragma("unroll") for (int j = 0; j < TP_MR; ++j) {                                              \
              const int r_ = r0_ + j;                                                                      \
              const bool on_ = (ok_) && r_ < r1_;                                                          \
              const int h_ = r_ >= RT ? 1 : 0;                                                             \
              const int u_ = r_ - h_ * RT;                                                                 \
              __amdgpu_buffer_rsrc_t rp_, rm_, rv2_;                                                       \
              TP_DESC(rp_, f.emb1, pi_, h_, on_); TP_DESC(rm_, f.m1, pi_, h_, on_); TP_DESC(rv2_, f.v1, pi_, h_, on_); \
              const int e_ = on_ ? (h_ * TI * H + 4 * at + 4 * TP_AT * u_) : 0;                            \
              const f32x4_t g4 = *(const f32x4_t*)(F_l + e_);                                              \
              f32x4_t p = P[set_][j], m = M[set_][j], v = V[set_][j];                                      \
              TP_ADAM1(p[0], m[0], v[0], g4[0]); TP_ADAM1(p[1], m[1], v[1], g4[1]);                        \
              TP_ADAM1(p[2], m[2], v[2], g4[2]); TP_ADAM1(p[3], m[3], v[3], g4[3]);                        \
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, p), rp_, vo, 2048 * u_, 2);   \
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, m), rm_, vo, 2048 * u_, 2);   \
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rv2_, vo, 2048 * u_, 2);  \
          } }
        // One chunk slot: rounds of slot s_ (pair it - 1), then the loads of the slot two ahead into the freed register set
#define TP_SLOT(set_, s_)                                                                                  \
        { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                  \
          TP_ROUNDS(set_, pprev, s_, it >= 1);                                                             \
          if ((s_) + 2 < nch) { TP_ISSUE(set_, pprev, (s_) + 2, it >= 1); }                                \
          else { TP_ISSUE(set_, pcur, (s_) + 2 - nch, it < n_it); } }
        for (int it = 0; it <= n_it; ++it) {
            const int pcur = (int)blockIdx.x + G * it;                      // the pair the GEMM waves work on (it < n_it)
            const int pprev = pcur - G;                                     // the pair whose dE is staged (it >= 1)
            const int* mc = meta_l + ((it + 1) & 1) * 4 * TM_LIST;          // list records of pprev
            int* mn = meta_l + (it & 1) * 4 * TM_LIST;                      // ... of pcur (written in slot 2)
            // ---------------- slot 0: sparse rows of pprev into the staged tile (thread t owns columns t and t + 128)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (it >= 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tl = a.tile_off + 2 * pprev + h;
                    const int tile0 = tl * TI;
                    const int id_lo = tile0 + 1, id_hi = (tl < a.tile_end ? min(tile0 + TI, N) : tile0) + 1;
                    float* Fh = F_l + h * TI * H;
                    const int* ms = mc + h * 2 * TM_LIST;
                    const int* mg = ms + TM_LIST;
                    const int k0s = ms[0], k1s = ms[1];
                    if (id_lo < id_hi) {
#pragma unroll
                        for (int i = 0; i < SPV; ++i) {
                            if (k0s + i < k1s) {
                                const int id = ms[2 + 2 * i];
                                if (id < id_hi) {
                                    Fh[(id - id_lo) * H + col0] += spv[h][i][0];
                                    if (has1) Fh[(id - id_lo) * H + col1] += spv[h][i][1];
                                }
                            }
                        }
                        for (int k = k0s + SPV; k < k1s; ++k) {              // (entries beyond the prefetched three: ~9 % of the tiles)
                            const int i = k - k0s;
                            const int id = i < 8 ? ms[2 + 2 * i] : f.sp_ids[k];
                            const int rw = i < 8 ? ms[3 + 2 * i] : f.sp_rows[k];
                            if (id < id_hi) {
                                const float v0 = f.sp_src[(size_t)rw * H + col0] * f.sp_scale;
                                const float v1 = has1 ? f.sp_src[(size_t)rw * H + col1] * f.sp_scale : 0.0f;
                                Fh[(id - id_lo) * H + col0] += v0;
                                if (has1) Fh[(id - id_lo) * H + col1] += v1;
                            }
                        }
                        for (int k = mg[0], k1 = mg[1]; k < k1; ++k) {       // one-hot target rows: dE[label] -= w_b rep_b
                            const int i = k - mg[0];
                            const int id = i < 8 ? mg[2 + 2 * i] : f.tg_ids[k];
                            const int bw = i < 8 ? mg[3 + 2 * i] : f.tg_rows[k];
                            if (id < id_hi) {
                                const float w = f.wrow[bw];
                                const float r0 = (float)a.rep_hi[(size_t)bw * LDR + col0] + (float)a.rep_lo[(size_t)bw * LDR + col0];
                                Fh[(id - id_lo) * H + col0] -= r0 * w * 1.0f;
                                if (has1) {
                                    const float r1 = (float)a.rep_hi[(size_t)bw * LDR + col1] + (float)a.rep_lo[(size_t)bw * LDR + col1];
                                    Fh[(id - id_lo) * H + col1] -= r1 * w * 1.0f;
                                }
                            }
                        }
                    }
                }
            }
            // the list records of pcur: one int per thread now, written to LDS in slot 2, read in slot 3
            {
                const bool on = it < n_it && at < 4 * TM_LIST && (a.tile_off + 2 * pcur + at / (2 * TM_LIST)) < a.tile_end;
                mreg = f.tile_meta[on ? (size_t)(a.tile_off + 2 * pcur) * (2 * TM_LIST) + at : 0];
                if (!on) mreg = 0;
            }
            TP_ISSUE(0, pprev, 2, it >= 1);
            // ---------------- slot 1
            TP_SLOT(1, 1);
            // ---------------- slot 2
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TP_ROUNDS(0, pprev, 2, it >= 1);
            if (at < 4 * TM_LIST) mn[at] = mreg;
            TP_ISSUE(0, pprev, 4, it >= 1);
            // ---------------- slot 3: the first input-embedding gradient rows of pcur's tiles (unconditional loads: row 0, column 0
            // where there is no entry), used in slot 0 of the next iteration
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TP_ROUNDS(1, pprev, 3, it >= 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int* ms = mn + h * 2 * TM_LIST;
#pragma unroll
                for (int i = 0; i < SPV; ++i) {
                    const bool on = ms[0] + i < ms[1];
                    const size_t rw = on ? (size_t)ms[3 + 2 * i] * H : 0;
                    spv[h][i][0] = f.sp_src[rw + (on ? col0 : 0)] * (on ? f.sp_scale : 0.0f);
                    spv[h][i][1] = f.sp_src[rw + ((on && has1) ? col1 : 0)] * ((on && has1) ? f.sp_scale : 0.0f);
                }
            }
            TP_ISSUE(1, pprev, 5, it >= 1);
            // ---------------- slots 4 .. nch - 1 (nch is a multiple of 4)
            for (int s = 4; s < nch; s += 2) {
                TP_SLOT(0, s);
                TP_SLOT(1, s + 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // H1
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // H2
        }
#undef TP_SLOT
#undef TP_ROUNDS
#undef TP_ISSUE
#undef TP_CUM
#undef TP_DESC
    }
}

static size_t tabp_lds(int Bp) {
    return (size_t)TP_NBUF * X3_IMG_B + (size_t)2 * TI * 150 * sizeof(float) + (size_t)Bp * sizeof(float) + 8 * TM_LIST * sizeof(int);
}

// Launch the pipelined kernel over tiles [a.tile_off, a.tile_off + tiles) if the shape is one it is built for; returns 1 if it
// was launched, 0 if the caller should use k_tab32x3 / k_tab16x3, < 0 / a hipError_t on failure.
int tabp_try_launch(TabArgs a, const FuseArgs& fa, int tiles, hipStream_t st) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -3;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const int nch = a.Bp / X3_CH;
    const int npairs = (tiles + 1) / 2;
    const int RT = (TI * a.H + 4 * TP_AT - 1) / (4 * TP_AT);
    // every CU gets >= 4 pairs (persistent workgroups: a short list would leave the pipeline mostly filling and draining); the
    // Adam rounds of a pair fit the chunk slots at TP_MR per slot; even first tile
    if (npairs < 4 * cus || (a.tile_off & 1) || nch < 6 || 2 * RT > TP_MR * (nch - 1) || a.H > 150 || a.Bp > 4096) return 0;
    a.tile_end = a.tile_off + tiles;
    const size_t lds = tabp_lds(a.Bp);
    static int lds_set = 0;
    if ((int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_tabp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = (int)lds;
    }
    hipLaunchKernelGGL(k_tabp, dim3(cus), dim3(512), lds, st, a, fa);
    return 1;
}
