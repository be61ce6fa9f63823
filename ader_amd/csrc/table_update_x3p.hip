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
//   waves 4-5  LOADER  copy the rep chunk images memory -> registers -> a ring of THREE LDS buffers, one chunk ahead in LDS and a
//                      second one in flight in their registers.
//   waves 6-7  ADAM    the previous pair: sparse input-embedding / one-hot rows into the staged dE tile, then TF-Adam over the pair's
//                      theta / m / v in float4 rounds spread evenly over the chunk slots of the GEMM waves' current pair, each
//                      round's loads requested two slots ahead -- the HBM stream runs at a constant rate under the matrix work.
// All eight waves meet at one s_barrier per chunk (+ two per pair around the dE hand-off).  Used for large catalogs (every CU gets
// several pairs); small ones, distilled steps and the EXTRA form keep k_tab32x3.  gfx950 only.
#include <stdlib.h>
#include "lbf_common.h"
#include "x3_image.h"
#include "../../include/ader_hip.h"

#define TI 64
#define X3_CH 32
#define TM_LIST 18
#define SPV 3
#define TP_NBUF 3                  // rep chunk buffers
#define TP_MR 3                    // Adam rounds (128 threads x float4) per chunk slot, at most
#define TP_D 4                     // ... and their loads are requested this many slots ahead (a ring of TP_D register sets)
#define TP_AT 128                  // Adam threads

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

#ifdef TP_STAMP     // diagnostic build only (tools/build_variant.sh ... -DTP_STAMP): clocks per segment of one wave of each role
__device__ unsigned long long tp_dbg[16 * 256];
#define TPS_INIT unsigned long long seg[4] = {0, 0, 0, 0}, tprev; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#define TPS(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                  __builtin_amdgcn_sched_barrier(0); seg[k_] += t_ - tprev; tprev = t_; }
#define TPS_DUMP(base_) { if (lane == 0) for (int k_ = 0; k_ < 4; ++k_) tp_dbg[blockIdx.x * 16 + (base_) + k_] = seg[k_]; }
extern "C" int ader_dbg_read_tp(void* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tp_dbg), (size_t)n * 8); }
#else
#define TPS_INIT
#define TPS(k_)
#define TPS_DUMP(base_) {}
#endif
#if defined(TP_STAMP) && defined(TP_STAMP2)     // GEMM role: hand-off breakdown instead of the chunk-loop segments
#define TPG(k_)
#define TPH(k_) TPS(k_)
#else
#define TPG(k_) TPS(k_)
#define TPH(k_)
#endif

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
        // One wave per SIMD has to keep the matrix pipe busy on its own, so the chunk is software-pipelined by one phase: slot c runs
        // the S = rep.E^T products of chunk c and then the dE += P^T.rep products of chunk c - 1, with the exp2 / hi-lo split of chunk
        // c's S (vector work: ~400 issue cycles) placed piece by piece in the gaps of those MFMAs (an MFMA of this shape holds the
        // vector issue for 8 of its 16 cycles).  Chunks c and c - 1 sit in two of the three ring buffers; the loader fills the third.
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
        // operand reads of a chunk image at Bh_ (k_tab32x3's: row reads two k-steps ahead, transposed reads three blocks ahead)
#define X3_LOADA(set_, Bh_, ks_)                                                                          \
        { const char* ap_ = (Bh_) + a_off + X3_QUAD * (ks_);                                              \
          set_[0] = *(const bf16x8*)ap_; set_[1] = *(const bf16x8*)(ap_ + X3_PLANE_B);                    \
          set_[2] = *(const bf16x8*)(ap_ + 256); set_[3] = *(const bf16x8*)(ap_ + X3_PLANE_B + 256); }
#define X3_LOADT(set_, Bh_, cb_)                                                                          \
        { const bf16* tp_ = (const bf16*)((Bh_) + t_off + X3_QUAD * ((cb_) >> 1) + 512 * ((cb_) & 1));     \
          set_[0] = tr_read(tp_); set_[1] = tr_read(tp_ + 128);                                           \
          set_[2] = tr_read(tp_ + X3_PLANE_B / 2); set_[3] = tr_read(tp_ + X3_PLANE_B / 2 + 128); }
        // S of the chunk in buffer Bh_ (both tiles of the pair, two 16-row blocks): 60 MFMAs
#define TP_S_PHASE(Bh_)                                                                                    \
        { _Pragma("unroll") for (int h = 0; h < 2; ++h) { S[h][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; S[h][1] = (f32x4v){0.f, 0.f, 0.f, 0.f}; } \
          bf16x8 fa[2][4];                                                                                 \
          X3_LOADA(fa[0], Bh_, 0);                                                                         \
          X3_LOADA(fa[1], Bh_, 1);                                                                         \
          __builtin_amdgcn_sched_barrier(0);                                                               \
          _Pragma("unroll") for (int ks = 0; ks < 5; ++ks) {                                               \
              bf16x8* A_ = fa[ks & 1];                                                                     \
              _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                              \
                  S[h][0] = mfma16_bf16(A_[1], e_hi[h][ks], S[h][0]);                                      \
                  S[h][1] = mfma16_bf16(A_[3], e_hi[h][ks], S[h][1]);                                      \
                  S[h][0] = mfma16_bf16(A_[0], e_lo[h][ks], S[h][0]);                                      \
                  S[h][1] = mfma16_bf16(A_[2], e_lo[h][ks], S[h][1]);                                      \
                  S[h][0] = mfma16_bf16(A_[0], e_hi[h][ks], S[h][0]);                                      \
                  S[h][1] = mfma16_bf16(A_[2], e_hi[h][ks], S[h][1]);                                      \
              }                                                                                            \
              if (ks + 2 < 5) X3_LOADA(fa[ks & 1], Bh_, ks + 2);                                           \
              __builtin_amdgcn_sched_barrier(0);                                                           \
          } }
        // p = w_b softmax = exp2(S log2e + off_b) of one PAIR of values (tile h, row block rb, rows j0, j0 + 1) of the chunk whose offsets
        // are in oa, split into the hi / lo bf16 words of the NEXT dE phase's fragments (k order of a fragment: rows 4g..4g+3 of block
        // 0, then of block 1; one 32-bit word = two consecutive rows).  Ten instructions in six pieces of <= 8 issue cycles, one piece per
        // MFMA gap (an MFMA of this shape holds the vector issue for 8 of its 16 cycles: a piece that fits the other 8 is ~free).
        // Volatile asm ON PURPOSE: plain exp2 / casts have no ordering against sched_barrier -- instruction selection hoisted all
        // sixteen exps in front of the phase's first MFMA.  v_cvt_pk_bf16_f32 is the instruction hipcc emits for the (bf16) casts of
        // k_tab32x3: the same roundings.
#define TP_PQ(q2_) constexpr int h_ = (q2_) >> 2, rb_ = ((q2_) >> 1) & 1, j0_ = 2 * ((q2_) & 1), w_ = 2 * rb_ + ((q2_) & 1);
#define TP_G1(q2_) { TP_PQ(q2_) asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %6" : "=&v"(xt0), "=&v"(xt1)         \
                       : "v"(S[h_][rb_][j0_]), "v"(S[h_][rb_][j0_ + 1]), "s"(LOG2E), "v"(oa[rb_][j0_]), "v"(oa[rb_][j0_ + 1])); (void)w_; }
#define TP_G2(q2_) { asm volatile("v_exp_f32 %0, %0" : "+v"(xt0)); }
#define TP_G3(q2_) { asm volatile("v_exp_f32 %0, %0" : "+v"(xt1)); }
#define TP_G4(q2_) { TP_PQ(q2_) asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_lshlrev_b32 %1, 16, %0" : "=&v"(pn_h[h_][w_]), "=&v"(xf0) \
                       : "v"(xt0), "v"(xt1)); (void)j0_; }
#define TP_G5(q2_) { TP_PQ(q2_) asm volatile("v_and_b32 %0, 0xffff0000, %2\n\tv_sub_f32 %1, %3, %4" : "=&v"(xf1), "=&v"(xl0)             \
                       : "v"(pn_h[h_][w_]), "v"(xt0), "v"(xf0)); (void)j0_; }
#define TP_G6(q2_) { TP_PQ(q2_) asm volatile("v_sub_f32 %1, %2, %3\n\tv_cvt_pk_bf16_f32 %0, %4, %1" : "=&v"(pn_l[h_][w_]), "=&v"(xl1)     \
                       : "v"(xt1), "v"(xf1), "v"(xl0)); (void)j0_; }
#define TP_EXPPAIR(q2_) TP_G1(q2_) TP_G2(q2_) TP_G3(q2_) TP_G4(q2_) TP_G5(q2_) TP_G6(q2_)
#define TP_FRAG(x_) __builtin_bit_cast(bf16x8, (u32x4_t){x_[0], x_[1], x_[2], x_[3]})
        // dE += P^T.rep of the chunk in buffer Bh_ with the fragments pc_h / pc_l: 60 MFMAs; WITH_EXP: the exp pieces of the chunk just
        // multiplied into S ride in their gaps, one value pair per channel block (cb = 0 .. 7)
#define TP_GAP(WITH_EXP, PIECE, cb_)                                                                       \
        if (WITH_EXP && (cb_) < 8) { __builtin_amdgcn_sched_barrier(0);                                    \
            switch (cb_) { case 0: PIECE(0) break; case 1: PIECE(1) break; case 2: PIECE(2) break; case 3: PIECE(3) break; \
                           case 4: PIECE(4) break; case 5: PIECE(5) break; case 6: PIECE(6) break; default: PIECE(7) break; } \
            __builtin_amdgcn_sched_barrier(0); }
#define TP_DE_PHASE(Bh_, WITH_EXP)                                                                         \
        { bf16x4 ft[3][4];                                                                                 \
          const bf16x8 ch0 = TP_FRAG(pc_h[0]), ch1 = TP_FRAG(pc_h[1]), cl0 = TP_FRAG(pc_l[0]), cl1 = TP_FRAG(pc_l[1]); \
          X3_LOADT(ft[0], Bh_, 0);                                                                         \
          X3_LOADT(ft[1], Bh_, 1);                                                                         \
          X3_LOADT(ft[2], Bh_, 2);                                                                         \
          __builtin_amdgcn_sched_barrier(0);                                                               \
          _Pragma("unroll") for (int cb = 0; cb < 10; ++cb) {                                              \
              bf16x4* T_ = ft[cb % 3];                                                                     \
              bf16x8 bh, bl;                                                                               \
              _Pragma("unroll") for (int j = 0; j < 4; ++j) { bh[j] = T_[0][j]; bh[4 + j] = T_[1][j]; bl[j] = T_[2][j]; bl[4 + j] = T_[3][j]; } \
              dE[0][cb] = mfma16_bf16(cl0, bh, dE[0][cb]);                                                 \
              TP_GAP(WITH_EXP, TP_G1, cb)                                                                  \
              dE[1][cb] = mfma16_bf16(cl1, bh, dE[1][cb]);                                                 \
              TP_GAP(WITH_EXP, TP_G2, cb)                                                                  \
              dE[0][cb] = mfma16_bf16(ch0, bl, dE[0][cb]);                                                 \
              TP_GAP(WITH_EXP, TP_G3, cb)                                                                  \
              dE[1][cb] = mfma16_bf16(ch1, bl, dE[1][cb]);                                                 \
              TP_GAP(WITH_EXP, TP_G4, cb)                                                                  \
              dE[0][cb] = mfma16_bf16(ch0, bh, dE[0][cb]);                                                 \
              TP_GAP(WITH_EXP, TP_G5, cb)                                                                  \
              dE[1][cb] = mfma16_bf16(ch1, bh, dE[1][cb]);                                                 \
              TP_GAP(WITH_EXP, TP_G6, cb)                                                                  \
              if (cb + 3 < 10) X3_LOADT(ft[cb % 3], Bh_, cb + 3);                                          \
              __builtin_amdgcn_sched_barrier(0);                                                           \
          } }
#define TP_LOAD_OFF(c_)                                                                                    \
        { const float4 o0_ = *(const float4*)(off_l + (c_) * X3_CH + 4 * g);                               \
          const float4 o1_ = *(const float4*)(off_l + (c_) * X3_CH + 16 + 4 * g);                          \
          oa[0][0] = o0_.x; oa[0][1] = o0_.y; oa[0][2] = o0_.z; oa[0][3] = o0_.w;                          \
          oa[1][0] = o1_.x; oa[1][1] = o1_.y; oa[1][2] = o1_.z; oa[1][3] = o1_.w; }
        TP_LOAD_X((int)blockIdx.x);
        TP_CUT_X();
        f32x4v dE[2][10];
        f32x4v S[2][2];
        unsigned pc_h[2][4], pc_l[2][4], pn_h[2][4], pn_l[2][4];            // hi / lo fragments of the dE phase: current chunk, next chunk
        float xt0, xt1, xf0, xf1, xl0, xl1;                                 // temporaries of the exp pieces
        float oa[2][4];
        int kb = 0;                                                         // ring buffer of the chunk about to be read
        TPS_INIT
        for (int it = 0; it <= n_it; ++it) {
            const bool act = it < n_it && !(a.ko & 2);          // (ko: timing-only knock-outs of ADER_DIAG builds; 0 otherwise)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int cb = 0; cb < 10; ++cb) dE[h][cb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            // ---- slot 0: S of chunk 0, its exp section in the open (once per pair)
            TPG(3)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TPG(0)
            const char* Bprev = (const char*)(R_l + kb * X3_BUF);
            if (act) {
                TP_LOAD_OFF(0);
                TP_S_PHASE(Bprev);
                TP_EXPPAIR(0) TP_EXPPAIR(1) TP_EXPPAIR(2) TP_EXPPAIR(3) TP_EXPPAIR(4) TP_EXPPAIR(5) TP_EXPPAIR(6) TP_EXPPAIR(7)
            }
            kb = (kb == TP_NBUF - 1) ? 0 : kb + 1;
            // ---- slots 1 .. nch - 1: S of chunk c, then dE of chunk c - 1 with the exp section of chunk c inside it
            for (int c = 1; c < nch; ++c) {
                TPG(2)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                TPG(0)
                const char* Bh = (const char*)(R_l + kb * X3_BUF);
                if (act) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int w = 0; w < 4; ++w) { pc_h[h][w] = pn_h[h][w]; pc_l[h][w] = pn_l[h][w]; }
                    TP_LOAD_OFF(c);
                    TP_S_PHASE(Bh);
                    TPG(1)
                    TP_DE_PHASE(Bprev, true);
                }
                Bprev = Bh;
                kb = (kb == TP_NBUF - 1) ? 0 : kb + 1;
            }
            // ---- dE of the last chunk (its buffer stays untouched until the loader's next-but-one fill, two barriers away)
            if (act) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int w = 0; w < 4; ++w) { pc_h[h][w] = pn_h[h][w]; pc_l[h][w] = pn_l[h][w]; }
                TP_DE_PHASE(Bprev, false);
            }
            TPG(2)
#if defined(TP_STAMP) && defined(TP_STAMP2)
            { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); tprev = t_; }
#endif
            // the next pair's theta rows (operand cut), requested before the hand-off so that their latency runs under it
            // (unconditional: beyond the workgroup's last pair the descriptor has zero records -- a load under a branch would be
            //  waited for at the end of that branch)
            TP_LOAD_X((int)blockIdx.x + G * (it + 1));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // H1: the Adam waves are done with the staged tile
            TPH(0)
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
            TPH(1)
#if defined(TP_STAMP) && defined(TP_STAMP2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            TPH(2)
            TP_CUT_X();
            TPH(3)
        }
        TPG(3)
        if (wave == 0) TPS_DUMP(0)
#undef TP_LOAD_OFF
#undef TP_DE_PHASE
#undef TP_GAP
#undef TP_FRAG
#undef TP_EXPPAIR
#undef TP_G1
#undef TP_G2
#undef TP_G3
#undef TP_G4
#undef TP_G5
#undef TP_G6
#undef TP_PQ
#undef TP_S_PHASE
#undef X3_LOADA
#undef X3_LOADT
#undef TP_LOAD_X
#undef TP_CUT_X
    } else if (wave < 6) {
        // =========================================================================================== loader waves
        // A chunk image (22 KiB, already in LDS layout) travels memory -> registers -> LDS: 11 sixteen-byte pieces per lane and wave.
        // (First build: LDS-DMA, two chunks ahead.  A global_load_lds costs its wave ~165 cycles of issue -- two waves x 11 pieces
        // set a floor of 1.37 us per slot, above the matrix time of a chunk; plain loads issue in a few cycles each, and with the
        // registers of a dedicated wave the copy costs 11 loads + 11 ds_write_b128.)
        const int dw = wave - 4;
        const int total = n_it * nch;                                       // chunks the GEMM waves will read
        const int lo = 11264 * dw + 16 * lane;                              // this lane's first byte of an image
        u32x4_t r[11];
#define TP_FETCH(chunk_)                                                                                   \
        { const char* src_ = img + (size_t)(chunk_) * X3_IMG_B + lo;                                       \
          _Pragma("unroll") for (int i = 0; i < 11; ++i) r[i] = *(const u32x4_t*)(src_ + 1024 * i); }
#define TP_PUT(buf_)                                                                                       \
        { char* dst_ = (char*)(R_l + (buf_) * X3_BUF) + lo;                                                \
          _Pragma("unroll") for (int i = 0; i < 11; ++i) *(u32x4_t*)(dst_ + 1024 * i) = r[i]; }
        TP_FETCH(0);
        TP_PUT(0);
        TP_FETCH(1 % nch);
        int k = 0, kb1 = 1, ch2 = 2 % nch;                                  // slot counter; buffer of chunk k + 1; image of chunk k + 2
        TPS_INIT
        for (int it = 0; it <= n_it; ++it) {
            for (int c = 0; c < nch; ++c) {
                TPS(2)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                TPS(0)
                // behind barrier k the GEMM waves are done with chunk k - 2: its buffer takes chunk k + 1 (in registers since the
                // previous slot), and chunk k + 2 is requested
                if (k + 1 < total) TP_PUT(kb1);
                TPS(1)
                TP_FETCH(ch2);
                ++k;
                kb1 = (kb1 == TP_NBUF - 1) ? 0 : kb1 + 1;
                ch2 = (ch2 == nch - 1) ? 0 : ch2 + 1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // H1
            asm volatile("s_barrier" ::: "memory");                            // H2
        }
        if (wave == 4) TPS_DUMP(4)
#undef TP_FETCH
#undef TP_PUT
    } else {
        // =========================================================================================== Adam waves
        const int at = tid - 384;                                           // 0..127
        const int R = (2 * TI * H + 4 * TP_AT - 1) / (4 * TP_AT);           // float4 rounds of 128 threads over a pair's 128 * H floats
        const int col0 = at, col1 = at + TP_AT;
        const bool has1 = col1 < H;
        f32x4_t P[TP_D][TP_MR], M[TP_D][TP_MR], V[TP_D][TP_MR];
        float spv[2][SPV][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < SPV; ++i) { spv[h][i][0] = 0.0f; spv[h][i][1] = 0.0f; }
        int mreg = 0;
        const int vo = 16 * at;
        // A pair's two tiles are ONE contiguous block of 128 * H floats in theta / m / v (and in F_l): the Adam rounds walk it flat,
        // round r = float4 index 4 at + 512 r of the block, r = 0 .. R - 1.  One buffer descriptor per array and pair (num_records =
        // the block's valid bytes: 0 if the pair is absent, less than 128 rows in the table's last pair); a round that does not exist
        // in a slot is given an offset beyond every block, so the hardware range check drops its loads (zeros) and stores -- no
        // branch around any memory instruction, and next to no scalar work per round.  (First builds: per-round descriptor selection
        // and integer divisions -- ~280 scalar instructions per slot; the Adam waves needed 2,700-4,800 clocks per slot and paced the
        // whole kernel, stamps in DESIGN.md.)
        __amdgpu_buffer_rsrc_t dP[2], dM[2], dV[2];                         // [0] the staged pair (pprev), [1] the GEMM waves' pair (pcur)
#define TP_PAIR_SETUP(w_, pi_, ok_)                                                                        \
        { const int tl_ = a.tile_off + 2 * (pi_);                                                          \
          const int tile0_ = tl_ * TI;                                                                     \
          const int cap_ = (tl_ + 1 < a.tile_end) ? 2 * TI : TI;                                           \
          const int rv_ = ((ok_) && tl_ < a.tile_end && !(a.ko & 1)) ? min(cap_, N - tile0_) : 0;          \
          const unsigned nb_ = rv_ > 0 ? (unsigned)(rv_ * H) * 4u : 0u;                                     \
          const size_t o_ = (size_t)(nb_ ? tile0_ : 0) * H;                                                \
          dP[w_] = __builtin_amdgcn_make_buffer_rsrc((void*)(f.emb1 + o_), 0, nb_, 0x00020000);            \
          dM[w_] = __builtin_amdgcn_make_buffer_rsrc((void*)(f.m1 + o_), 0, nb_, 0x00020000);              \
          dV[w_] = __builtin_amdgcn_make_buffer_rsrc((void*)(f.v1 + o_), 0, nb_, 0x00020000); }
        // slot s (1 <= s <= nch - 1) takes rounds [floor((s - 1) R / den), floor(s R / den)), kept incrementally: rem = s R mod den
        const int den = nch - 1;
        int p_r = 0, p_rem = 0;                                             // cursor of the slot being processed
        int i_r = 0, i_rem = 0, i_slot = TP_D % nch;                        // cursor TP_D slots ahead: next round, target slot
        __amdgpu_buffer_rsrc_t iP, iM, iV;                                  // ... and the descriptors of the pair it is in
#define TP_ADVANCE(r_, rem_, cnt_)                                                                         \
        { rem_ += R; cnt_ = 0;                                                                             \
          _Pragma("unroll") for (int k_ = 0; k_ < TP_MR; ++k_) if (rem_ >= den) { rem_ -= den; ++cnt_; } }
#define TP_OOB 0x40000000
        // issue the loads of the target slot (TP_D ahead) into register set set_, then move the issue cursor one slot on
#define TP_ISSUE_AHEAD(set_)                                                                               \
        { int cnt_ = 0;                                                                                    \
          if (i_slot != 0) { TP_ADVANCE(i_r, i_rem, cnt_) }                                                \
          _Pragma("unroll") for (int j = 0; j < TP_MR; ++j) {                                              \
              const int so_ = j < cnt_ ? 2048 * (i_r + j) : TP_OOB;                                        \
              P[set_][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(iP, vo, so_, 0));   \
              M[set_][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(iM, vo, so_, 0));   \
              V[set_][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(iV, vo, so_, 0));   \
          }                                                                                                \
          i_r += cnt_;                                                                                     \
          if (++i_slot == nch) { i_slot = 0; i_r = 0; i_rem = 0; iP = dP[1]; iM = dM[1]; iV = dV[1]; } }
        // Adam on the rounds of the current slot (the staged pair) held in register set set_
#define TP_ROUNDS(set_, s_)                                                                                \
        { int cnt_ = 0;                                                                                    \
          if ((s_) != 0) { TP_ADVANCE(p_r, p_rem, cnt_) }                                                  \
          _Pragma("unroll") for (int j = 0; j < TP_MR; ++j) {                                              \
              const bool on_ = j < cnt_;                                                                   \
              const int so_ = on_ ? 2048 * (p_r + j) : TP_OOB;                                             \
              const int e_ = on_ ? 4 * at + 4 * TP_AT * (p_r + j) : 0;                                     \
              const f32x4_t g4 = *(const f32x4_t*)(F_l + e_);                                              \
              f32x4_t p = P[set_][j], m = M[set_][j], v = V[set_][j];                                      \
              TP_ADAM1(p[0], m[0], v[0], g4[0]); TP_ADAM1(p[1], m[1], v[1], g4[1]);                        \
              TP_ADAM1(p[2], m[2], v[2], g4[2]); TP_ADAM1(p[3], m[3], v[3], g4[3]);                        \
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, p), dP[0], vo, so_, 2);   \
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, m), dM[0], vo, so_, 2);   \
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), dV[0], vo, so_, 2);   \
          }                                                                                                \
          p_r += cnt_; }
        // One chunk slot: Adam on the rounds of slot s_ (pair it - 1), then the loads TP_D slots ahead.  The HBM latency under load is
        // 4-5 us and a slot lasts ~1.3 us: with two slots of lookahead (first build) the slot time settled at latency / 2 and the
        // whole kernel ran at the Adam waves' pace (1.30 ms against k_tab32x3's 0.99)
#define TP_SLOT(set_, s_)                                                                                  \
        { TPS(1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); TPS(0)                      \
          TP_ROUNDS(set_, s_);                                                                             \
          if ((s_) >= 1 && (s_) <= 10) TP_TOUCH(s_)                                                        \
          TP_ISSUE_AHEAD(set_); }
        // The GEMM waves cut their operands for pair pcur + G from that pair's theta rows at the end of this iteration, with nothing to
        // hide the HBM latency behind (stamps: the hand-off took 9,600 clocks per pair): the Adam waves TOUCH those 77 KB during the
        // iteration -- one dword per 64 bytes, 4 KB per wave-instruction, one instruction per wave and slot, result never read -- so
        // that the rows wait in L2 / Infinity Cache.  (The destination register is a dedicated one, alive to the end of the kernel:
        // the loads land in it whenever they complete.)
        unsigned pf = 0;
        u32x4_t dX = {0u, 0u, 0u, 0u};
#define TP_TOUCH(s_)                                                                                       \
        { const unsigned vo_ = (unsigned)(64 * lane);                                                      \
          const unsigned so_ = (unsigned)(8192 * ((s_) - 1) + 4096 * (wave - 6));                           \
          asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(pf) : "v"(vo_), "s"(dX), "s"(so_) : "memory"); }
        // prologue: no pair is staged in iteration 0 (absent: zero records); the issue cursor starts at slot TP_D of that iteration
        TP_PAIR_SETUP(0, 0, false);
        TP_PAIR_SETUP(1, 0, false);
        iP = dP[0]; iM = dM[0]; iV = dV[0];
        TPS_INIT
        for (int it = 0; it <= n_it; ++it) {
            const int pcur = (int)blockIdx.x + G * it;                      // the pair the GEMM waves work on (it < n_it)
            const int pprev = pcur - G;                                     // the pair whose dE is staged (it >= 1)
            const int* mc = meta_l + ((it + 1) & 1) * 4 * TM_LIST;          // list records of pprev
            int* mn = meta_l + (it & 1) * 4 * TM_LIST;                      // ... of pcur (written in slot 2)
            // what was the GEMM waves' pair is now the staged one ([1] -> [0]; the issue cursor has been reading it as [1] since it
            // crossed into this iteration's slots and goes on with the same descriptors)
            dP[0] = dP[1]; dM[0] = dM[1]; dV[0] = dV[1];
            iP = dP[0]; iM = dM[0]; iV = dV[0];
            TP_PAIR_SETUP(1, pcur, it < n_it);
            {   // theta block of the pair after pcur (zero records if this workgroup has none)
                const int tl_ = a.tile_off + 2 * (pcur + G);
                const int tile0_ = tl_ * TI;
                const int cap_ = (tl_ + 1 < a.tile_end) ? 2 * TI : TI;
                const int rv_ = (it + 1 < n_it && tl_ < a.tile_end) ? min(cap_, a.vrows - tile0_) : 0;
                const unsigned nb_ = rv_ > 0 ? (unsigned)(rv_ * H) * 4u : 0u;
                const unsigned long long pa_ = (unsigned long long)(uintptr_t)(a.emb1 + (size_t)(nb_ ? tile0_ : 0) * H);
                dX = (u32x4_t){(unsigned)pa_, (unsigned)(pa_ >> 32) & 0xffffu, nb_, 0x00020000u};   // raw buffer V#: base, stride 0, records, flags
            }
            p_r = 0; p_rem = 0;
            // ---------------- slot 0: sparse rows of pprev into the staged tile (thread t owns columns t and t + 128)
            TPS(1)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TPS(0)
            if (it >= 1) {
                // (no mul + add contraction in the sparse rows: every x3 update kernel forms  row * scale  and  F + that  as two
                //  rounded operations, so that they agree bit for bit whatever the compiler would fuse in each of them)
#pragma clang fp contract(off)
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
            TPS(2)
            TP_ISSUE_AHEAD(0);
            // ---------------- slot 1
            TP_SLOT(1, 1);
            // ---------------- slot 2
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TP_ROUNDS(2, 2);
            TP_TOUCH(2)
            if (at < 4 * TM_LIST) mn[at] = mreg;
            TP_ISSUE_AHEAD(2);
            // ---------------- slot 3: the first input-embedding gradient rows of pcur's tiles (unconditional loads: row 0, column 0
            // where there is no entry), used in slot 0 of the next iteration
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TP_ROUNDS(3, 3);
            TP_TOUCH(3)
            {
#pragma clang fp contract(off)
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
            }
            TP_ISSUE_AHEAD(3);
            // ---------------- slots 4 .. nch - 1 (nch is a multiple of 4)
            for (int s = 4; s < nch; s += 4) {
                TP_SLOT(0, s);
                TP_SLOT(1, s + 1);
                TP_SLOT(2, s + 2);
                TP_SLOT(3, s + 3);
            }
            TPS(1)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // H1
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // H2
            TPS(3)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" :: "v"(pf));
        if (wave == 6) TPS_DUMP(8)
#undef TP_TOUCH
#undef TP_SLOT
#undef TP_ISSUE_AHEAD
#undef TP_ROUNDS
#undef TP_OOB
#undef TP_ADVANCE
#undef TP_PAIR_SETUP
    }
}

static size_t tabp_lds(int Bp) {
    return (size_t)TP_NBUF * X3_IMG_B + (size_t)2 * TI * 150 * sizeof(float) + (size_t)Bp * sizeof(float) + 8 * TM_LIST * sizeof(int);
}

// Launch the pipelined kernel over tiles [a.tile_off, a.tile_off + tiles) if the shape is one it is built for; returns 1 if it
// was launched, 0 if the caller should use k_tab32x3 / k_tab16x3, < 0 / a hipError_t on failure.
int tabp_try_launch(TabArgs a, const FuseArgs& fa, int tiles, int wg_per_cu, hipStream_t st) {
    static int cus_dev[ADER_MAX_DEV] = {};
    int& cus = cus_dev[ader_cur_dev()];
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -3;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const int nch = a.Bp / X3_CH;
    const int npairs = (tiles + 1) / 2;
    const int R2 = (2 * TI * a.H + 4 * TP_AT - 1) / (4 * TP_AT);      // Adam rounds per pair
    // every CU gets >= 4 pairs (persistent workgroups: a short list would leave the pipeline mostly filling and draining); the
    // Adam rounds of a pair fit the chunk slots at TP_MR per slot; even first tile
    const int G = cus * (wg_per_cu < 1 ? 1 : wg_per_cu);
    if (npairs < 4 * G || (a.tile_off & 1) || nch < 2 * TP_D || (nch % TP_D) || R2 > TP_MR * (nch - 1) || a.H > 150 || a.Bp > 4096) return 0;
    a.tile_end = a.tile_off + tiles;
    a.ko = 0;
#ifdef ADER_DIAG
    { static int ko = -1; if (ko < 0) { const char* e = getenv("ADER_TP_KO"); ko = e ? atoi(e) : 0; } a.ko = ko; }
#endif
    const size_t lds = tabp_lds(a.Bp);
    static int lds_set_dev[ADER_MAX_DEV] = {};
    int& lds_set = lds_set_dev[ader_cur_dev()];
    if ((int)lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_tabp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set = (int)lds;
    }
    hipLaunchKernelGGL(k_tabp, dim3(G), dim3(512), lds, st, a, fa);
    return 1;
}
