// Full-catalog logits + softmax cross-entropy at float32 grade: the flash forward kernels on the conflict-free block images.
// Reference: ADER.py:88-93 (logits = rep . item_emb^T, one-hot softmax CE) in float32: every product as three bf16 MFMAs on hi/lo
// operand splits (hi.hi + lo.hi + hi.lo, ~2^-16 relative, fp32 accumulate, fp32 softmax).
//
// Same algorithm as k_lx3_fwd (logits_bf16.hip): per 32-item block S^T = E.rep^T, online max / sum-exp, and the probabilities go
// back into the matrix core as the A operand of O[b,:] += P^T.E (the softmax-weighted readout = dRep up to the target term), so
// nothing [rows, N]-sized is ever written.  The table block is split into hi/lo on its way into LDS as the bank-conflict-free image
// of x3_image.h (16-byte k-chunks [kc][item][8 channels]), and both operand reads -- ds_read_b128 rows for S^T, ds_read_b64_tr_b16
// k-major for the readout -- are pipelined by hand ahead of their MFMAs.
//   k_lx3g  32 batch rows per wave on v_mfma_f32_32x32x16_bf16 (128 per workgroup, two workgroups per CU); any supported H
//   k_lx3p  k_lx3g with the softmax / staging vector work issued inside the MFMA phases (H = 150: the default forward)
//   k_lx3r  the teacher readout of distilled steps (ADER.py:132-137) on the same images
// (k_lx3f, 16 rows per wave on 16x16x32 tiles, and k_lx3h, k_lx3g's blocking on 16x16x32 tiles, were measured 25 % and 4 % slower in
// round 3 -- DESIGN.md 6a -- and removed in round 4.)
// Output partials (pm, pl, pO per item range) and the merge (k_lbf_combine<true>) are those of k_lx3_fwd.  gfx950 only.
#include "lbf_common.h"
#include "x3_image.h"
#include "../../include/ader_hip.h"

#define F3_FB 32                   // items per streamed block
#define F3_RND 3                   // staging rounds: 12 units of (8 items x 8 k-chunks), 4 waves

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(8))) F3Vec { f32x4_t v; };      // 16-byte vector at an 8-byte aligned address (rows are 8 H bytes)

// ---------------------------------------------------------------------------------------------------------------------------
// k_lx3g: 32 batch rows per wave on v_mfma_f32_32x32x16_bf16 (128 per workgroup, two workgroups per CU): every LDS operand fragment
// feeds twice the flops of a 16x16x32 form with 16-row waves, which keeps the LDS pipe as busy as the matrix pipe.
// Register plan (<= 256): rep fragments 80, O 80, S / P 16, one block in flight 24, operand sets 16 / 32.
#define G3_ROWS 128
#ifdef G3_STAMP     // diagnostic build only (tools/build_variant.sh ... -DG3_STAMP): per-segment clocks of wave 0 of every workgroup
__device__ unsigned long long g3_dbg[8 * 1024];
#define STAMP(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                    __builtin_amdgcn_sched_barrier(0); seg[k_] += t_ - tprev; tprev = t_; }
extern "C" int ader_dbg_read(void* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g3_dbg), (size_t)n * 8); }
#else
#define STAMP(k_)
#endif
template <int HT>
__global__ __launch_bounds__(256, 2) void k_lx3g(Lx3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];      // [2 buffers][block image]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int nchunk = a.Bp / G3_ROWS;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range = xcd + 8 * (slot / nchunk);           // the row chunks of an item range sit on one XCD: the table block is
    const int bc = slot % nchunk;                          // fetched from HBM once and served to the others by that XCD's L2
    if (range >= a.ranges) return;
    const int H = HT ? HT : a.H;
    const int N = (bc * G3_ROWS >= a.kd_row0) ? a.Np : a.N;            // columns of this chunk's softmax (distilled rows: first Np)
    const int nblk_all = (a.N + F3_FB - 1) / F3_FB;
    const int per = (nblk_all + a.ranges - 1) / a.ranges;
    const int blk_begin = range * per, blk_end = min((N + F3_FB - 1) / F3_FB, blk_begin + per);
    const int nb_blocks = max(0, blk_end - blk_begin);
    const int b0 = bc * G3_ROWS + wave * 32;
    // pads of both images (k-chunks >= ceil(H/8), bytes between the quads) stay zero: the block stores never touch them
    for (int i = tid; i < 3 * X3B_IMG_B / 16; i += 256) ((uint4*)smem_raw)[i] = make_uint4(0u, 0u, 0u, 0u);
    // rep fragments: lane (batch row r32, k-half hh) holds rep[b0 + r32][16 ks + 8 hh + 0..7]
    bf16x8 rh[10], rl[10];
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        rh[ks] = *(const bf16x8*)(a.rep_hi + (size_t)(b0 + r32) * LDR + 16 * ks + 8 * hh);
        rl[ks] = *(const bf16x8*)(a.rep_lo + (size_t)(b0 + r32) * LDR + 16 * ks + 8 * hh);
    }
    f32x16 O[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    // ---- staging: a block = 32 table rows of H floats.  Round r (0..2) of wave w covers items 8 w.. and k-chunks 8 r..:
    // lane l -> item + (l & 7), k-chunk + (l >> 3): the 8 lanes of an LDS write group store 8 consecutive 16-byte
    // slots (conflict-free), and a wave's two 16-byte loads per slot touch 2 cache lines per table row.
    const int nfull = H >> 3, rem = H & 7;                 // full k-chunks; channels of the partial one (0, 4 or 6: see launcher)
    // round r of this lane: item it_ (= 8 wave + (lane & 7): a wave stages the same 8 items in every round), k-chunk 8 r + kc0
    const int it_ = 8 * wave + (lane & 7), kc0 = lane >> 3;
    const int voff = 4 * (it_ * H + 8 * kc0);              // byte offset inside the block (round r: + 256 r)
    const int voffp = voff + 4 * (rem - 4);                // second vector of the partial k-chunk: ends with the row
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)a.emb1, 0, a.vrows * H * 4, 0x00020000);
    const int dst0 = X3B_KC * kc0 + 16 * it_;               // byte offset inside a plane (round r: + 8 k-chunks)
    // ---- H = 150: the block as 16-byte pieces read in memory order.  Lane l -> item 8 wave + (l >> 3), piece (l & 7) + 8 r of the
    // row's 37.5 (round r = 0..4): a wave-instruction reads 8 x 128 contiguous bytes -- ~12 cache lines, every byte used -- where the
    // k-chunk mapping above reads 8 x 8 half-used 32-byte pieces (~20 lines, each touched by two instructions); 5 loads instead of
    // 6.  A piece = 4 channels = half a k-chunk slot: one ds_write_b64 per plane (two-way bank conflicts, hidden under the
    // VGPR-to-LDS transfer of the store).
    constexpr bool MAPB = (HT == 150);
    const int itb = 8 * wave + (lane >> 3), qb = lane & 7;
    const int voffb = itb * (4 * 150) + 16 * qb;           // round r: + 128 r
    const int dstb = X3B_KC * (qb >> 1) + 16 * itb + 8 * (qb & 1);      // round r: + 4 k-chunks
    f32x4_t sc[5];
#define G3B_LOAD(blk_)                                                                                    \
    {                                                                                                     \
        const int so_ = (blk_) * (F3_FB * 4) * 150;                                                       \
        _Pragma("unroll") for (int r = 0; r < 5; ++r)                                                     \
            sc[r] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(trs, voffb + 128 * r, so_, 0)); \
    }
#define G3B_STORE(buf_)                                                                                   \
    {                                                                                                     \
        unsigned char* dst_ = smem_raw + (buf_) * X3B_IMG_B;                                              \
        _Pragma("unroll") for (int r = 0; r < 5; ++r) {                                                   \
            const int p_ = qb + 8 * r;                          /* piece 37 = channels 148, 149 and two floats of the next row */ \
            float x_[4];                                        /* pieces 38, 39 = channels 152..159: zeros (no branch) */ \
            x_[0] = (r == 4 && p_ >= 38) ? 0.f : sc[r][0]; x_[1] = (r == 4 && p_ >= 38) ? 0.f : sc[r][1]; \
            x_[2] = (r == 4 && p_ >= 37) ? 0.f : sc[r][2]; x_[3] = (r == 4 && p_ >= 37) ? 0.f : sc[r][3]; \
            bf16x4 h_, l_;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { h_[j] = (bf16)x_[j]; l_[j] = (bf16)(x_[j] - (float)h_[j]); } \
            *(bf16x4*)(dst_ + dstb + 4 * X3B_KC * r) = h_;                                                \
            *(bf16x4*)(dst_ + X3B_PLANE_B + dstb + 4 * X3B_KC * r) = l_;                                  \
        }                                                                                                 \
    }
#define G3_LOADBLK(blk_) { if constexpr (MAPB) G3B_LOAD(blk_) else F3_LOAD(blk_) }
#define G3_STOREBLK(buf_) { if constexpr (MAPB) G3B_STORE(buf_) else F3_STORE(buf_) }
#define F3_KC(r_) (8 * (r_) + kc0)
#define F3_PART(r_) (rem && F3_KC(r_) == nfull)
#define F3_VALID(r_) (F3_KC(r_) < nfull || F3_PART(r_))
    f32x4_t sa[F3_RND], sb[F3_RND];
    // Block loads through a buffer descriptor of the table (base in scalar registers, ONE 32-bit per-lane offset, the block's
    // offset as the scalar offset, the round's as the instruction's immediate): no 64-bit per-lane pointers, and rows beyond the
    // table's last one (only in its last block; their items are >= N: outside the softmax) come back as zeros from the hardware
    // range check.  Lanes without a k-chunk read whatever follows their row (never stored).  NOTHING is selected on the loaded data
    // here -- a select would make hipcc wait for each load right behind its issue.
#define F3_LOAD(blk_)                                                                                     \
    {                                                                                                     \
        const int so_ = (blk_) * (F3_FB * 4) * H;              /* byte offset of the block (< 2^31: checked by the launcher) */ \
        _Pragma("unroll") for (int r = 0; r < F3_RND; ++r) {                                              \
            const u32x4_t va_ = __builtin_amdgcn_raw_buffer_load_b128(trs, voff + 256 * r, so_, 0);         \
            const u32x4_t vb_ = __builtin_amdgcn_raw_buffer_load_b128(trs, (F3_PART(r) ? voffp : voff + 16) + 256 * r, so_, 0); \
            sa[r] = __builtin_bit_cast(f32x4_t, va_); sb[r] = __builtin_bit_cast(f32x4_t, vb_);           \
        }                                                                                                 \
    }
    // hi = bf16(x), lo = bf16(x - hi), 8 channels -> one 16-byte slot per plane
#define F3_STORE(buf_)                                                                                    \
    {                                                                                                     \
        unsigned char* dst_ = smem_raw + (buf_) * X3B_IMG_B;                                               \
        _Pragma("unroll") for (int r = 0; r < F3_RND; ++r) {                                              \
            float x_[8];                                                                                  \
            x_[0] = sa[r][0]; x_[1] = sa[r][1]; x_[2] = sa[r][2]; x_[3] = sa[r][3];                       \
            if (F3_PART(r)) {       /* rem = 6: channels 4,5 are elements 2,3 of the shifted vector; rem = 4: none */ \
                x_[4] = (rem == 6) ? sb[r][2] : 0.f; x_[5] = (rem == 6) ? sb[r][3] : 0.f; x_[6] = 0.f; x_[7] = 0.f; \
            } else { x_[4] = sb[r][0]; x_[5] = sb[r][1]; x_[6] = sb[r][2]; x_[7] = sb[r][3]; }            \
            bf16x8 h_, l_;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { h_[j] = (bf16)x_[j]; l_[j] = (bf16)(x_[j] - (float)h_[j]); } \
            if (F3_VALID(r)) {                                                                            \
                *(bf16x8*)(dst_ + dst0 + 8 * X3B_KC * r) = h_;                                            \
                *(bf16x8*)(dst_ + X3B_PLANE_B + dst0 + 8 * X3B_KC * r) = l_;                              \
            }                                                                                             \
        }                                                                                                 \
    }
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    // per-lane byte offsets into a block image: row read of (item r32, k-half hh); transposed read of (item 4 hh + q4, channels
    // 16 g1 + 4 p4.. of a 32-channel block = k-chunks 2 g1 + (p4 >> 1) of its four)
    const int a_off = X3B_KC * hh + 16 * r32;
    const int t_off = X3B_KC * (2 * g1 + (p4 >> 1)) + 16 * (4 * hh + q4) + 8 * (p4 & 1);
    // three LDS buffers: block i is read from buffer i % 3 while block i + 1 (stored during iteration i - 1) waits in the next one and
    // block i + 2 -- requested at the head of iteration i, converted and stored between its two MFMA phases -- goes into the third:
    // the 24 staging registers are live only under the S^T phase, where the operand sets are small
    if (nb_blocks > 0) G3_LOADBLK(blk_begin);
    __syncthreads();                                       // zero fill done
    if (nb_blocks > 0) G3_STOREBLK(0);
    if (nb_blocks > 1) { G3_LOADBLK(blk_begin + 1); G3_STOREBLK(1); }
    int bcur = 0;                                          // i % 3
#ifdef G3_STAMP
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#endif
    for (int i = 0; i < nb_blocks; ++i) {
        STAMP(0)
        __syncthreads();                                   // blocks i and i + 1 are in LDS; every wave is done with block i - 1
        STAMP(1)
        const bool more = i + 2 < nb_blocks;
        if (more) G3_LOADBLK(blk_begin + i + 2);
        STAMP(2)
        const char* Bh = (const char*)(smem_raw + bcur * X3B_IMG_B);
        const int bnew = bcur == 0 ? 2 : bcur - 1;         // (i + 2) % 3
        const int i0 = (blk_begin + i) * F3_FB;
        // S^T = 32 items x 32 batch rows: A = table rows (lane: item r32, k = 8 hh..8 hh + 7 of the k-step), B = rep fragments
#define G3_LOADA(set_, ks_)                                                                               \
        { const char* ap_ = Bh + a_off + 2 * X3B_KC * (ks_);                                              \
          set_[0] = *(const bf16x8*)ap_; set_[1] = *(const bf16x8*)(ap_ + X3B_PLANE_B); }
        // transposed reads of the 32-channel block nb: {hi: items 4hh.., 8 + 4hh.., 16 + 4hh.., 24 + 4hh..; lo: the same}
#define G3_LOADT(set_, nb_, pl_)                                                                          \
        { const bf16* tp_ = (const bf16*)(Bh + t_off + 4 * X3B_KC * (nb_) + (pl_) * X3B_PLANE_B);            \
          set_[0] = tr_read(tp_); set_[1] = tr_read(tp_ + 64); set_[2] = tr_read(tp_ + 128); set_[3] = tr_read(tp_ + 192); }
        f32x16 S;
#pragma unroll
        for (int j = 0; j < 16; ++j) S[j] = 0.0f;
        bf16x8 fa[2][2];
        G3_LOADA(fa[0], 0);
        G3_LOADA(fa[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            bf16x8* A_ = fa[ks & 1];                             // {hi, lo}
            S = mfma_bf16(A_[1], rh[ks], S);
            S = mfma_bf16(A_[0], rl[ks], S);
            S = mfma_bf16(A_[0], rh[ks], S);
            if (ks + 2 < 10) G3_LOADA(fa[ks & 1], ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(3)
        if (more) G3_STOREBLK(bnew);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4)
        // the first transposed reads of the readout do not depend on S: in flight under the softmax section
        bf16x4 ft[3][4];        // half sets: {items 4hh.., 8 + 4hh.., 16 + 4hh.., 24 + 4hh..} of ONE plane; hi, lo, hi, lo ...
        G3_LOADT(ft[0], 0, 0);
        G3_LOADT(ft[1], 0, 1);
        G3_LOADT(ft[2], 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i0 + F3_FB > N) {                              // tail block: items >= N are outside the softmax
#pragma unroll
            for (int j = 0; j < 16; ++j) if (i0 + acc_row(j, hh) >= N) S[j] = -INFINITY;
        }
        float tmax = S[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, S[j]);
        // the two half-waves of a lane pair (l, l + 32) hold different items of the SAME batch row; m_run is kept equal in both, so
        // the cross-half exchange is only needed on the (rare) rescale path
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
        bf16x8 pl0, pl1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pl0[j] = (bf16)(S[j] - (float)pa0[j]); pl1[j] = (bf16)(S[8 + j] - (float)pa1[j]); }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5)
#pragma unroll
        for (int hs = 0; hs < 10; ++hs) {                        // half step: (channel block nb = hs >> 1, plane hs & 1)
            bf16x4* T_ = ft[hs % 3];
            bf16x8 v0, v1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { v0[j] = T_[0][j]; v0[4 + j] = T_[1][j]; v1[j] = T_[2][j]; v1[4 + j] = T_[3][j]; }
            if ((hs & 1) == 0) {                                 // hi plane of the table: P lo and P hi
                O[hs >> 1] = mfma_bf16(pl0, v0, O[hs >> 1]);
                O[hs >> 1] = mfma_bf16(pl1, v1, O[hs >> 1]);
                O[hs >> 1] = mfma_bf16(pa0, v0, O[hs >> 1]);
                O[hs >> 1] = mfma_bf16(pa1, v1, O[hs >> 1]);
            } else {                                             // lo plane: P hi
                O[hs >> 1] = mfma_bf16(pa0, v0, O[hs >> 1]);
                O[hs >> 1] = mfma_bf16(pa1, v1, O[hs >> 1]);
            }
            if (hs + 3 < 10) G3_LOADT(ft[hs % 3], (hs + 3) >> 1, (hs + 3) & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        bcur = bcur == 2 ? 0 : bcur + 1;
        STAMP(6)
    }
#ifdef G3_STAMP
    if (tid == 0 && blockIdx.x < 1024) { for (int k_ = 0; k_ < 8; ++k_) g3_dbg[blockIdx.x * 8 + k_] = seg[k_]; g3_dbg[blockIdx.x * 8 + 7] = nb_blocks; }
#endif
#undef G3_LOADA
#undef G3_LOADT
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (hh == 0) {
        a.pm[(size_t)range * a.Bp + b0 + r32] = m_run;
        a.pl[(size_t)range * a.Bp + b0 + r32] = l_tot;
    }
    float* o = a.pO + ((size_t)range * a.Bp + b0) * HP;
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[(size_t)acc_row(j, hh) * HP + 32 * nb + r32] = O[nb][j];
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_lx3p: k_lx3g with the softmax of block i issued INSIDE the S phase of block i + 1.  Stamps of k_lx3g: a wave
// spends 2,400 of a block's 5,300 clocks in vector-only phases (staging, softmax) and the two waves of a SIMD mostly take turns --
// the kernel is bound by the waves' serial chains, not by the matrix pipe (68 % busy).  Here the logits of the NEXT block are
// accumulated (a second S, 16 registers) while the exp / sum / hi-lo split of the current block's logits are placed between its
// MFMAs; the rescale test (a branch) stays in front.  Same arithmetic in the same order as k_lx3g: bit-equal results.
template <int HT>
__global__ __launch_bounds__(256, 2) void k_lx3p(Lx3Args a) {
    // (product build: LOG2E_S = LOG2E and the two factors are 1: the compiler folds them; -DADER_X3_F16: see lbf_common.h)
    constexpr float LOG2E_S = LOG2E / (X3_SR * X3_SE);
    constexpr float X3_INV_P = 1.0f / (X3_SPL == 0.0f ? 1.0f : 256.0f), X3_INV_PE = X3_INV_P / X3_SE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];      // [2 buffers][block image]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int nchunk = a.Bp / G3_ROWS;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range = xcd + 8 * (slot / nchunk);           // the row chunks of an item range sit on one XCD: the table block is
    const int bc = slot % nchunk;                          // fetched from HBM once and served to the others by that XCD's L2
    if (range >= a.ranges) return;
    const int H = HT ? HT : a.H;
    const int N = (bc * G3_ROWS >= a.kd_row0) ? a.Np : a.N;            // columns of this chunk's softmax (distilled rows: first Np)
    const int nblk_all = (a.N + F3_FB - 1) / F3_FB;
    const int per = (nblk_all + a.ranges - 1) / a.ranges;
    const int blk_begin = range * per, blk_end = min((N + F3_FB - 1) / F3_FB, blk_begin + per);
    const int nb_blocks = max(0, blk_end - blk_begin);
    const int b0 = bc * G3_ROWS + wave * 32;
    // pads of both images (k-chunks >= ceil(H/8), bytes between the quads) stay zero: the block stores never touch them
    for (int i = tid; i < 3 * X3B_IMG_B / 16; i += 256) ((uint4*)smem_raw)[i] = make_uint4(0u, 0u, 0u, 0u);
    // rep fragments: lane (batch row r32, k-half hh) holds rep[b0 + r32][16 ks + 8 hh + 0..7]
    bf16x8 rh[10], rl[10];
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        rh[ks] = *(const bf16x8*)(a.rep_hi + (size_t)(b0 + r32) * LDR + 16 * ks + 8 * hh);
        rl[ks] = *(const bf16x8*)(a.rep_lo + (size_t)(b0 + r32) * LDR + 16 * ks + 8 * hh);
    }
    f32x16 O[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    // ---- staging: a block = 32 table rows of H floats.  Round r (0..2) of wave w covers items 8 w.. and k-chunks 8 r..:
    // lane l -> item + (l & 7), k-chunk + (l >> 3): the 8 lanes of an LDS write group store 8 consecutive 16-byte
    // slots (conflict-free), and a wave's two 16-byte loads per slot touch 2 cache lines per table row.
    const int nfull = H >> 3, rem = H & 7;                 // full k-chunks; channels of the partial one (0, 4 or 6: see launcher)
    // round r of this lane: item it_ (= 8 wave + (lane & 7): a wave stages the same 8 items in every round), k-chunk 8 r + kc0
    const int it_ = 8 * wave + (lane & 7), kc0 = lane >> 3;
    const int voff = 4 * (it_ * H + 8 * kc0);              // byte offset inside the block (round r: + 256 r)
    const int voffp = voff + 4 * (rem - 4);                // second vector of the partial k-chunk: ends with the row
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)a.emb1, 0, a.vrows * H * 4, 0x00020000);
    const int dst0 = X3B_KC * kc0 + 16 * it_;               // byte offset inside a plane (round r: + 8 k-chunks)
    // ---- H = 150: the block as 16-byte pieces read in memory order.  Lane l -> item 8 wave + (l >> 3), piece (l & 7) + 8 r of the
    // row's 37.5 (round r = 0..4): a wave-instruction reads 8 x 128 contiguous bytes -- ~12 cache lines, every byte used -- where the
    // k-chunk mapping above reads 8 x 8 half-used 32-byte pieces (~20 lines, each touched by two instructions); 5 loads instead of
    // 6.  A piece = 4 channels = half a k-chunk slot: one ds_write_b64 per plane (two-way bank conflicts, hidden under the
    // VGPR-to-LDS transfer of the store).
    constexpr bool MAPB = (HT == 150);
    const int itb = 8 * wave + (lane >> 3), qb = lane & 7;
    const int voffb = itb * (4 * 150) + 16 * qb;           // round r: + 128 r
    const int dstb = X3B_KC * (qb >> 1) + 16 * itb + 8 * (qb & 1);      // round r: + 4 k-chunks
    f32x4_t sc[5];
#define G3B_LOAD(blk_)                                                                                    \
    {                                                                                                     \
        const int so_ = (blk_) * (F3_FB * 4) * 150;                                                       \
        _Pragma("unroll") for (int r = 0; r < 5; ++r)                                                     \
            sc[r] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(trs, voffb + 128 * r, so_, 0)); \
    }
#define G3B_STORE(buf_)                                                                                   \
    {                                                                                                     \
        unsigned char* dst_ = smem_raw + (buf_) * X3B_IMG_B;                                              \
        _Pragma("unroll") for (int r = 0; r < 5; ++r) {                                                   \
            const int p_ = qb + 8 * r;                          /* piece 37 = channels 148, 149 and two floats of the next row */ \
            float x_[4];                                        /* pieces 38, 39 = channels 152..159: zeros (no branch) */ \
            x_[0] = (r == 4 && p_ >= 38) ? 0.f : sc[r][0] * X3_SE; x_[1] = (r == 4 && p_ >= 38) ? 0.f : sc[r][1] * X3_SE; \
            x_[2] = (r == 4 && p_ >= 37) ? 0.f : sc[r][2] * X3_SE; x_[3] = (r == 4 && p_ >= 37) ? 0.f : sc[r][3] * X3_SE; \
            bf16x4 h_, l_;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { h_[j] = (bf16)x_[j]; l_[j] = (bf16)(x_[j] - (float)h_[j]); } \
            *(bf16x4*)(dst_ + dstb + 4 * X3B_KC * r) = h_;                                                \
            *(bf16x4*)(dst_ + X3B_PLANE_B + dstb + 4 * X3B_KC * r) = l_;                                  \
        }                                                                                                 \
    }
#define G3_LOADBLK(blk_) { if constexpr (MAPB) G3B_LOAD(blk_) else F3_LOAD(blk_) }
#define G3_STOREBLK(buf_) { if constexpr (MAPB) G3B_STORE(buf_) else F3_STORE(buf_) }
#define F3_KC(r_) (8 * (r_) + kc0)
#define F3_PART(r_) (rem && F3_KC(r_) == nfull)
#define F3_VALID(r_) (F3_KC(r_) < nfull || F3_PART(r_))
    f32x4_t sa[F3_RND], sb[F3_RND];
    // Block loads through a buffer descriptor of the table (base in scalar registers, ONE 32-bit per-lane offset, the block's
    // offset as the scalar offset, the round's as the instruction's immediate): no 64-bit per-lane pointers, and rows beyond the
    // table's last one (only in its last block; their items are >= N: outside the softmax) come back as zeros from the hardware
    // range check.  Lanes without a k-chunk read whatever follows their row (never stored).  NOTHING is selected on the loaded data
    // here -- a select would make hipcc wait for each load right behind its issue.
#define F3_LOAD(blk_)                                                                                     \
    {                                                                                                     \
        const int so_ = (blk_) * (F3_FB * 4) * H;              /* byte offset of the block (< 2^31: checked by the launcher) */ \
        _Pragma("unroll") for (int r = 0; r < F3_RND; ++r) {                                              \
            const u32x4_t va_ = __builtin_amdgcn_raw_buffer_load_b128(trs, voff + 256 * r, so_, 0);         \
            const u32x4_t vb_ = __builtin_amdgcn_raw_buffer_load_b128(trs, (F3_PART(r) ? voffp : voff + 16) + 256 * r, so_, 0); \
            sa[r] = __builtin_bit_cast(f32x4_t, va_); sb[r] = __builtin_bit_cast(f32x4_t, vb_);           \
        }                                                                                                 \
    }
    // hi = bf16(x), lo = bf16(x - hi), 8 channels -> one 16-byte slot per plane
#define F3_STORE(buf_)                                                                                    \
    {                                                                                                     \
        unsigned char* dst_ = smem_raw + (buf_) * X3B_IMG_B;                                               \
        _Pragma("unroll") for (int r = 0; r < F3_RND; ++r) {                                              \
            float x_[8];                                                                                  \
            x_[0] = sa[r][0]; x_[1] = sa[r][1]; x_[2] = sa[r][2]; x_[3] = sa[r][3];                       \
            if (F3_PART(r)) {       /* rem = 6: channels 4,5 are elements 2,3 of the shifted vector; rem = 4: none */ \
                x_[4] = (rem == 6) ? sb[r][2] : 0.f; x_[5] = (rem == 6) ? sb[r][3] : 0.f; x_[6] = 0.f; x_[7] = 0.f; \
            } else { x_[4] = sb[r][0]; x_[5] = sb[r][1]; x_[6] = sb[r][2]; x_[7] = sb[r][3]; }            \
            bf16x8 h_, l_;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) { h_[j] = (bf16)x_[j]; l_[j] = (bf16)(x_[j] - (float)h_[j]); } \
            if (F3_VALID(r)) {                                                                            \
                *(bf16x8*)(dst_ + dst0 + 8 * X3B_KC * r) = h_;                                            \
                *(bf16x8*)(dst_ + X3B_PLANE_B + dst0 + 8 * X3B_KC * r) = l_;                              \
            }                                                                                             \
        }                                                                                                 \
    }
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    // per-lane byte offsets into a block image: row read of (item r32, k-half hh); transposed read of (item 4 hh + q4, channels
    // 16 g1 + 4 p4.. of a 32-channel block = k-chunks 2 g1 + (p4 >> 1) of its four)
    const int a_off = X3B_KC * hh + 16 * r32;
    const int t_off = X3B_KC * (2 * g1 + (p4 >> 1)) + 16 * (4 * hh + q4) + 8 * (p4 & 1);
    // three LDS buffers: block i is read from buffer i % 3 while block i + 1 (stored during iteration i - 1) waits in the next one and
    // block i + 2 -- requested at the head of iteration i, converted and stored between its two MFMA phases -- goes into the third:
    // the 24 staging registers are live only under the S^T phase, where the operand sets are small
    if (nb_blocks > 0) G3_LOADBLK(blk_begin);
    __syncthreads();                                       // zero fill done
    if (nb_blocks > 0) G3_STOREBLK(0);
    if (nb_blocks > 1) { G3_LOADBLK(blk_begin + 1); G3_STOREBLK(1); }
    int bcur = 0;                                          // i % 3
    // S^T = 32 items x 32 batch rows: A = table rows (lane: item r32, k = 8 hh..8 hh + 7 of the k-step), B = rep fragments
#define G3_LOADA(set_, ks_)                                                                               \
    { const char* ap_ = Bs + a_off + 2 * X3B_KC * (ks_);                                                  \
      set_[0] = *(const bf16x8*)ap_; set_[1] = *(const bf16x8*)(ap_ + X3B_PLANE_B); }
#define G3_LOADT(set_, nb_, pl_)                                                                          \
    { const bf16* tp_ = (const bf16*)(Bh + t_off + 4 * X3B_KC * (nb_) + (pl_) * X3B_PLANE_B);              \
      set_[0] = tr_read(tp_); set_[1] = tr_read(tp_ + 64); set_[2] = tr_read(tp_ + 128); set_[3] = tr_read(tp_ + 192); }
    // softmax of ONE pair of logits of the current block (elements 2 q_, 2 q_ + 1 of S): p = exp2(s log2e - m), running sum in
    // element order (the order of k_lx3g: bit-equal results), hi / lo split into packed pairs
#define P3_PAIR(q_)                                                                                       \
    { const float p0_ = __builtin_amdgcn_exp2f(fmaf(S[2 * (q_)], LOG2E_S, nm));                             \
      const float p1_ = __builtin_amdgcn_exp2f(fmaf(S[2 * (q_) + 1], LOG2E_S, nm));                         \
      ls += p0_; ls += p1_;                                                                               \
      bf16x2 h_; h_[0] = (bf16)p0_; h_[1] = (bf16)p1_;                                                    \
      bf16x2 l_; l_[0] = (bf16)(p0_ - (float)h_[0]); l_[1] = (bf16)(p1_ - (float)h_[1]);                  \
      ph2[q_] = __builtin_bit_cast(uint32_t, h_); pl2[q_] = __builtin_bit_cast(uint32_t, l_); }
    f32x16 S;
    // ---- block 0's logits: a plain S phase (no softmax to hide yet)
    __syncthreads();                                       // blocks 0 and 1 are in LDS
    if (nb_blocks > 0) {
        const char* Bs = (const char*)smem_raw;
#pragma unroll
        for (int j = 0; j < 16; ++j) S[j] = 0.0f;
        bf16x8 fa[2][2];
        G3_LOADA(fa[0], 0);
        G3_LOADA(fa[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            bf16x8* A_ = fa[ks & 1];
            S = mfma_bf16(A_[1], rh[ks], S);
            S = mfma_bf16(A_[0], rl[ks], S);
            S = mfma_bf16(A_[0], rh[ks], S);
            if (ks + 2 < 10) G3_LOADA(fa[ks & 1], ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the part of an iteration in front of the S phase / softmax: tail mask, running maximum, (rare) rescale of l and O
#define P3_HEAD()                                                                                         \
        const int i0 = (blk_begin + i) * F3_FB;                                                           \
        if (i0 + F3_FB > N) {                              /* tail block: items >= N are outside the softmax */ \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) if (i0 + acc_row(j, hh) >= N) S[j] = -INFINITY; \
        }                                                                                                 \
        float tmax = S[0];                                                                                \
        _Pragma("unroll") for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, S[j]);                          \
        float t2 = tmax * LOG2E_S;                                                                          \
        if (__any(t2 > m_run + RESCALE_THR)) {                                                            \
            t2 = fmaxf(t2, __shfl_xor(t2, 32, 64));                                                       \
            const float m_new = (t2 > m_run + RESCALE_THR) ? t2 : m_run;                                  \
            const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);      \
            l_run *= alpha;                                                                               \
            m_run = m_new;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                              \
                const float ar = __shfl(alpha, acc_row(j, hh), 64);     /* O rows are batch rows */       \
                _Pragma("unroll") for (int nb = 0; nb < 5; ++nb) O[nb][j] *= ar;                          \
            }                                                                                             \
        }                                                                                                 \
        const float nm = (X3_SPL == 0.0f) ? -m_run : X3_SPL - m_run;     /* (product build: -m_run) */            \
        float ls = 0.0f;                                                                                  \
        uint32_t ph2[8], pl2[8];                           /* P hi / lo as packed bf16 pairs */
    // ... and behind it: the block in flight goes to LDS, then the readout O += P^T . E of the current block
#define P3_TAIL()                                                                                         \
        l_run += ls;                                                                                      \
        if constexpr (more) G3_STOREBLK(bnew);                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        bf16x4 ft[3][4];                                                                                  \
        G3_LOADT(ft[0], 0, 0);                                                                            \
        G3_LOADT(ft[1], 0, 1);                                                                            \
        G3_LOADT(ft[2], 1, 0);                                                                            \
        bf16x8 pa0, pa1, pl0, pl1;                                                                        \
        {   typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));                                  \
            pa0 = __builtin_bit_cast(bf16x8, (u32x4_){ph2[0], ph2[1], ph2[2], ph2[3]});                   \
            pa1 = __builtin_bit_cast(bf16x8, (u32x4_){ph2[4], ph2[5], ph2[6], ph2[7]});                   \
            pl0 = __builtin_bit_cast(bf16x8, (u32x4_){pl2[0], pl2[1], pl2[2], pl2[3]});                   \
            pl1 = __builtin_bit_cast(bf16x8, (u32x4_){pl2[4], pl2[5], pl2[6], pl2[7]});                   \
        }                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        _Pragma("unroll") for (int hs = 0; hs < 10; ++hs) {      /* half step: (channel block nb = hs >> 1, plane hs & 1) */ \
            bf16x4* T_ = ft[hs % 3];                                                                      \
            bf16x8 v0, v1;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { v0[j] = T_[0][j]; v0[4 + j] = T_[1][j]; v1[j] = T_[2][j]; v1[4 + j] = T_[3][j]; } \
            if ((hs & 1) == 0) {                                 /* hi plane of the table: P lo and P hi */ \
                O[hs >> 1] = mfma_bf16(pl0, v0, O[hs >> 1]);                                              \
                O[hs >> 1] = mfma_bf16(pl1, v1, O[hs >> 1]);                                              \
                O[hs >> 1] = mfma_bf16(pa0, v0, O[hs >> 1]);                                              \
                O[hs >> 1] = mfma_bf16(pa1, v1, O[hs >> 1]);                                              \
            } else {                                             /* lo plane: P hi */                     \
                O[hs >> 1] = mfma_bf16(pa0, v0, O[hs >> 1]);                                              \
                O[hs >> 1] = mfma_bf16(pa1, v1, O[hs >> 1]);                                              \
            }                                                                                             \
            if (hs + 3 < 10) G3_LOADT(ft[hs % 3], (hs + 3) >> 1, (hs + 3) & 1);                           \
            __builtin_amdgcn_sched_barrier(0);                                                            \
        }
    int i = 0;
    for (; i + 2 < nb_blocks; ++i) {                       // (every iteration stages a block: NO branch between the S phase and the readout --
        __syncthreads();                                   //  across one the compiler sinks the whole softmax to its first use)
        constexpr bool more = true;
        G3_LOADBLK(blk_begin + i + 2);
        const int bnext = bcur == 2 ? 0 : bcur + 1;        // (i + 1) % 3
        const int bnew = bcur == 0 ? 2 : bcur - 1;         // (i + 2) % 3
        const char* Bh = (const char*)(smem_raw + bcur * X3B_IMG_B);
        const char* Bs = (const char*)(smem_raw + bnext * X3B_IMG_B);
        P3_HEAD()
        // ---- the NEXT block's S phase, with this block's softmax in the shadows of its MFMAs: a matrix instruction holds the SIMD's
        // vector issue for 8 of its 32 clocks, so a few vector instructions placed BETWEEN the three MFMAs of a k-step cost nothing
        // while the matrix pipe is busy.  Per k-step: the exp of one pair of logits, and the sum / hi-lo split of the PREVIOUS pair
        // (its exp results are a k-step old: no wait).  hipcc does not interleave the two chains by itself: the order is pinned as
        // written (a sched_barrier after every piece).
        f32x16 Sn;
#pragma unroll
        for (int j = 0; j < 16; ++j) Sn[j] = 0.0f;
        bf16x8 fa[2][2];
        G3_LOADA(fa[0], 0);
        G3_LOADA(fa[1], 1);
        __builtin_amdgcn_sched_barrier(0);
        // (order pinned piece by piece: the machine scheduler's own interleaving -- sched_group_barrier -- is reverted at 254
        //  registers, and vector instructions left to instruction selection all land behind the MFMAs)
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            bf16x8* A_ = fa[ks & 1];
            Sn = mfma_bf16(A_[1], rh[ks], Sn);
            __builtin_amdgcn_sched_barrier(0);
            if (ks < 8) {                                   // exp of pair ks
                S[2 * ks] = __builtin_amdgcn_exp2f(fmaf(S[2 * ks], LOG2E_S, nm));
                S[2 * ks + 1] = __builtin_amdgcn_exp2f(fmaf(S[2 * ks + 1], LOG2E_S, nm));
            }
            __builtin_amdgcn_sched_barrier(0);
            Sn = mfma_bf16(A_[0], rl[ks], Sn);
            __builtin_amdgcn_sched_barrier(0);
            if (ks >= 1 && ks < 9) {                        // pair ks - 1 (its exp results are a k-step old): sum, hi
                const int q = ks - 1;
                ls += S[2 * q]; ls += S[2 * q + 1];
                bf16x2 h_; h_[0] = (bf16)S[2 * q]; h_[1] = (bf16)S[2 * q + 1];
                ph2[q] = __builtin_bit_cast(uint32_t, h_);
                S[2 * q] -= (float)h_[0]; S[2 * q + 1] -= (float)h_[1];
            }
            __builtin_amdgcn_sched_barrier(0);
            Sn = mfma_bf16(A_[0], rh[ks], Sn);
            __builtin_amdgcn_sched_barrier(0);
            if (ks >= 1 && ks < 9) {                        // ... lo
                const int q = ks - 1;
                bf16x2 l_; l_[0] = (bf16)S[2 * q]; l_[1] = (bf16)S[2 * q + 1];
                pl2[q] = __builtin_bit_cast(uint32_t, l_);
            }
            if (ks + 2 < 10) G3_LOADA(fa[ks & 1], ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        l_run += ls;
        // ---- the readout O += P^T . E of block i, with the conversion of block i + 2 (fp32 -> hi / lo image: ~17 vector instructions
        // and two 8-byte LDS stores per 16-byte piece) in the shadows of its MFMAs: round r of the staging goes with half steps 2 r, 2 r + 1
        static_assert(HT == 150, "k_lx3p: piece staging (H = 150) only");
        bf16x4 ft[3][4];
        G3_LOADT(ft[0], 0, 0);
        G3_LOADT(ft[1], 0, 1);
        G3_LOADT(ft[2], 1, 0);
        bf16x8 pa0, pa1, pl0, pl1;
        {   typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
            pa0 = __builtin_bit_cast(bf16x8, (u32x4_){ph2[0], ph2[1], ph2[2], ph2[3]});
            pa1 = __builtin_bit_cast(bf16x8, (u32x4_){ph2[4], ph2[5], ph2[6], ph2[7]});
            pl0 = __builtin_bit_cast(bf16x8, (u32x4_){pl2[0], pl2[1], pl2[2], pl2[3]});
            pl1 = __builtin_bit_cast(bf16x8, (u32x4_){pl2[4], pl2[5], pl2[6], pl2[7]});
        }
        unsigned char* dstn = smem_raw + bnew * X3B_IMG_B + dstb;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            float x_[4];
            bf16x4 h_, l_;
            // ---- half step 2 r: hi plane of the table (channel block r): P lo and P hi
            {
                bf16x4* T_ = ft[(2 * r) % 3];
                bf16x8 v0, v1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = T_[0][j]; v0[4 + j] = T_[1][j]; v1[j] = T_[2][j]; v1[4 + j] = T_[3][j]; }
                O[r] = mfma_bf16(pl0, v0, O[r]);
                __builtin_amdgcn_sched_barrier(0);
                {   const int p_ = qb + 8 * r;                  // pieces 38, 39 = channels 152..159: zeros (no branch)
                    x_[0] = (r == 4 && p_ >= 38) ? 0.f : sc[r][0] * X3_SE; x_[1] = (r == 4 && p_ >= 38) ? 0.f : sc[r][1] * X3_SE;
                    x_[2] = (r == 4 && p_ >= 37) ? 0.f : sc[r][2] * X3_SE; x_[3] = (r == 4 && p_ >= 37) ? 0.f : sc[r][3] * X3_SE;
                }
                __builtin_amdgcn_sched_barrier(0);
                O[r] = mfma_bf16(pl1, v1, O[r]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) h_[j] = (bf16)x_[j];
                __builtin_amdgcn_sched_barrier(0);
                O[r] = mfma_bf16(pa0, v0, O[r]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) x_[j] -= (float)h_[j];
                __builtin_amdgcn_sched_barrier(0);
                O[r] = mfma_bf16(pa1, v1, O[r]);
                if (2 * r + 3 < 10) G3_LOADT(ft[(2 * r) % 3], (2 * r + 3) >> 1, (2 * r + 3) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- half step 2 r + 1: lo plane: P hi
            {
                bf16x4* T_ = ft[(2 * r + 1) % 3];
                bf16x8 v0, v1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v0[j] = T_[0][j]; v0[4 + j] = T_[1][j]; v1[j] = T_[2][j]; v1[4 + j] = T_[3][j]; }
                O[r] = mfma_bf16(pa0, v0, O[r]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) l_[j] = (bf16)x_[j];
                *(bf16x4*)(dstn + 4 * X3B_KC * r) = h_;
                __builtin_amdgcn_sched_barrier(0);
                O[r] = mfma_bf16(pa1, v1, O[r]);
                __builtin_amdgcn_sched_barrier(0);
                *(bf16x4*)(dstn + X3B_PLANE_B + 4 * X3B_KC * r) = l_;
                if (2 * r + 4 < 10) G3_LOADT(ft[(2 * r + 1) % 3], (2 * r + 4) >> 1, (2 * r + 4) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        S = Sn;
        bcur = bnext;
    }
    for (; i < nb_blocks; ++i) {                           // ---- the last two blocks: plain order (nothing left to stage)
        __syncthreads();
        constexpr bool more = false;
        const int bnew = 0;
        const int bnext = bcur == 2 ? 0 : bcur + 1;
        const char* Bh = (const char*)(smem_raw + bcur * X3B_IMG_B);
        const char* Bs = (const char*)(smem_raw + bnext * X3B_IMG_B);
        P3_HEAD()
#pragma unroll
        for (int q = 0; q < 8; ++q) P3_PAIR(q);
        f32x16 Sn;
#pragma unroll
        for (int j = 0; j < 16; ++j) Sn[j] = 0.0f;
        if (i + 1 < nb_blocks) {
            bf16x8 fa[2][2];
            G3_LOADA(fa[0], 0);
            G3_LOADA(fa[1], 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                bf16x8* A_ = fa[ks & 1];
                Sn = mfma_bf16(A_[1], rh[ks], Sn);
                Sn = mfma_bf16(A_[0], rl[ks], Sn);
                Sn = mfma_bf16(A_[0], rh[ks], Sn);
                if (ks + 2 < 10) G3_LOADA(fa[ks & 1], ks + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        P3_TAIL()
        S = Sn;
        bcur = bnext;
    }
#undef P3_HEAD
#undef P3_TAIL
#undef P3_PAIR
#undef G3_LOADA
#undef G3_LOADT
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (hh == 0) {
        a.pm[(size_t)range * a.Bp + b0 + r32] = m_run;
        a.pl[(size_t)range * a.Bp + b0 + r32] = l_tot * X3_INV_P;
    }
    float* o = a.pO + ((size_t)range * a.Bp + b0) * HP;
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[(size_t)acc_row(j, hh) * HP + 32 * nb + r32] = O[nb][j] * X3_INV_PE;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_lx3r: the TEACHER READOUT of a distilled step (ADER.py:132-137) as its own launch: O2[b,:] = sum_{j < Np} softmax(teacher_b)_j E_j
// for the 128-row chunks of exemplar rows -- k_lx3g's readout half (same block images, same hand-pipelined transposed reads, same
// 30 MFMAs per block and wave) fed by the teacher logits instead of an S phase: lane (row r32, half hh) loads the 16 teacher
// logits of its accumulator positions (four 16-byte loads), p = exp2(t log2e - tlse2) is exact (the teacher's log-sum-exp is known),
// split hi / lo.  HBM-bound (teacher tile + table block per 32 items): both streams run TWO blocks ahead of their use in registers
// (table: two staging sets, converted into the third LDS buffer at the head of an iteration; teacher: two sets), and the block
// barrier orders LDS traffic only -- __syncthreads would drain those loads.  Replaces k_lx3_fwd<2, 2, true> (logits_bf16.hip: one
// block of lookahead, per-lane branches around the teacher loads; 0.30 ms against 0.2 at cfg-S + 128 exemplar rows) for H = 150,
// 16-byte aligned teacher rows.  Output partials pO2 [range][Bk][160] as before.
__global__ __launch_bounds__(256, 2) void k_lx3r(Lx3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];      // [3 buffers][block image]
    constexpr int H = 150;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    const int Bk = a.Bp - a.kd_row0;
    const int nchunk = Bk / G3_ROWS;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range = xcd + 8 * (slot / nchunk);
    const int bc = slot % nchunk;
    if (range >= a.ranges2) return;
    const int Np = a.Np;
    const int nblk_all = (Np + F3_FB - 1) / F3_FB;
    const int per = (nblk_all + a.ranges2 - 1) / a.ranges2;
    const int blk_begin = range * per, blk_end = min(nblk_all, blk_begin + per);
    const int nb_blocks = max(0, blk_end - blk_begin);
    const int bk0 = bc * G3_ROWS + wave * 32;              // first exemplar row of the wave (row of the padded batch: + kd_row0)
    for (int i = tid; i < 3 * X3B_IMG_B / 16; i += 256) ((uint4*)smem_raw)[i] = make_uint4(0u, 0u, 0u, 0u);
    f32x16 O[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    // table staging: k_lx3g's memory-order mapping (lane -> item 8 wave + (l >> 3), 16-byte piece (l & 7) + 8 r of the row)
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)a.emb1, 0, a.vrows * H * 4, 0x00020000);
    const int itb = 8 * wave + (lane >> 3), qb = lane & 7;
    const int voffb = itb * (4 * H) + 16 * qb;
    const int dstb = X3B_KC * (qb >> 1) + 16 * itb + 8 * (qb & 1);
    f32x4_t sc[2][5];
#define R3_LOAD(set_, blk_)                                                                               \
    {                                                                                                     \
        const int so_ = (blk_) * (F3_FB * 4) * H;                                                         \
        _Pragma("unroll") for (int r = 0; r < 5; ++r)                                                     \
            sc[set_][r] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(trs, voffb + 128 * r, so_, 0)); \
    }
#define R3_STORE(set_, buf_)                                                                              \
    {                                                                                                     \
        unsigned char* dst_ = smem_raw + (buf_) * X3B_IMG_B;                                              \
        _Pragma("unroll") for (int r = 0; r < 5; ++r) {                                                   \
            const int p_ = qb + 8 * r;                                                                    \
            float x_[4];                                                                                  \
            x_[0] = (r == 4 && p_ >= 38) ? 0.f : sc[set_][r][0]; x_[1] = (r == 4 && p_ >= 38) ? 0.f : sc[set_][r][1]; \
            x_[2] = (r == 4 && p_ >= 37) ? 0.f : sc[set_][r][2]; x_[3] = (r == 4 && p_ >= 37) ? 0.f : sc[set_][r][3]; \
            bf16x4 h_, l_;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { h_[j] = (bf16)x_[j]; l_[j] = (bf16)(x_[j] - (float)h_[j]); } \
            *(bf16x4*)(dst_ + dstb + 4 * X3B_KC * r) = h_;                                                \
            *(bf16x4*)(dst_ + X3B_PLANE_B + dstb + 4 * X3B_KC * r) = l_;                                  \
        }                                                                                                 \
    }
    // teacher logits of this lane's accumulator positions: row r32, items 8 q + 4 hh + 0..3 of the block (register 4 q + k)
    const int tr = a.trow[a.kd_row0 + bk0 + r32];
    const float tl2 = a.tlse2[a.kd_row0 + bk0 + r32];
    const float* trp = a.teacher + (size_t)(tr < 0 ? 0 : tr) * a.ldt + 4 * hh;
    float tt[2][16];
    // Everything inside the block loop is unconditional -- a load under a branch becomes a phi of two register sets (hipcc then spilled
    // a whole staging set: loads straight into scratch).  Table blocks past the range are real rows (or zeros from the descriptor's
    // range check) that are stored into a buffer nobody reads; teacher blocks are clamped to the last FULL block of [0, Np).  The one
    // partial block of a launch (Np % 32 != 0; last range only) is peeled: element-wise clamped loads, validity applied to p.
    const int nfull_all = Np / F3_FB;                      // >= 1 (launcher)
#define R3_TLOAD(set_, blk_)                                                                              \
    {                                                                                                     \
        const int i0_ = min((blk_), nfull_all - 1) * F3_FB;                                               \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
            const f32x4_t v_ = *(const f32x4_t*)(trp + i0_ + 8 * q);                                      \
            tt[set_][4 * q] = v_[0]; tt[set_][4 * q + 1] = v_[1]; tt[set_][4 * q + 2] = v_[2]; tt[set_][4 * q + 3] = v_[3]; \
        }                                                                                                 \
    }
#define R3_TSLOW(set_, blk_)                                                                              \
    {                                                                                                     \
        const int i0_ = (blk_) * F3_FB;                                                                   \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                     \
            _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                 \
                tt[set_][4 * q + k] = trp[min(i0_ + 8 * q + k, Np - 1 - 4 * hh)];                         \
    }
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int t_off = X3B_KC * (2 * g1 + (p4 >> 1)) + 16 * (4 * hh + q4) + 8 * (p4 & 1);
#define R3_LOADT(set_, nb_, pl_)                                                                          \
        { const bf16* tp_ = (const bf16*)(Bh + t_off + 4 * X3B_KC * (nb_) + (pl_) * X3B_PLANE_B);            \
          set_[0] = tr_read(tp_); set_[1] = tr_read(tp_ + 64); set_[2] = tr_read(tp_ + 128); set_[3] = tr_read(tp_ + 192); }
    const int nb_full = max(0, min(blk_end, nfull_all) - blk_begin);       // full blocks of this range (the rest: the partial one)
    // prologue: blocks 0, 1 -> LDS buffers 0, 1; blocks 2, 3 -> staging sets 0, 1; teacher blocks 0, 1 -> sets 0, 1
    R3_LOAD(0, blk_begin);
    R3_LOAD(1, blk_begin + 1);
    R3_TLOAD(0, blk_begin);
    R3_TLOAD(1, blk_begin + 1);
    __syncthreads();                                       // zero fill done
    R3_STORE(0, 0);
    R3_STORE(1, 1);
    R3_LOAD(0, blk_begin + 2);
    R3_LOAD(1, blk_begin + 3);
    int bcur = 0;                                          // i % 3
    // one iteration (block i, register sets e_ = i & 1): block i + 2 (requested two iterations ago) goes into the buffer block i - 1
    // was read from, block i + 4 is requested; P from the teacher set, which is then refilled with block i + 2
#define R3_ITER(e_)                                                                                       \
    {                                                                                                     \
        lds_only_barrier();                                /* blocks i, i + 1 in LDS; every wave is done with block i - 1 */ \
        const int bnew = bcur == 0 ? 2 : bcur - 1;         /* (i + 2) % 3 */                              \
        R3_STORE(e_, bnew);                                                                               \
        R3_LOAD(e_, blk_begin + i + 4);                                                                   \
        const char* Bh = (const char*)(smem_raw + bcur * X3B_IMG_B);                                      \
        bf16x4 ft[3][4];                                                                                  \
        R3_LOADT(ft[0], 0, 0);                                                                            \
        R3_LOADT(ft[1], 0, 1);                                                                            \
        R3_LOADT(ft[2], 1, 0);                                                                            \
        f32x16 S;                                                                                         \
        const int lim_ = (tr >= 0) ? Np - (blk_begin + i) * F3_FB - 4 * hh : 0;     /* items of this lane's positions inside [0, Np) */ \
        _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                  \
            const float x_ = __builtin_amdgcn_exp2f(fmaf(tt[e_][j], LOG2E, -tl2));                        \
            S[j] = ((j & 3) + 8 * (j >> 2) < lim_) ? x_ : 0.0f;                                           \
        }                                                                                                 \
        R3_TLOAD(e_, blk_begin + i + 2);                                                                  \
        const bf16x8 pa0 = pack8(S, 0), pa1 = pack8(S, 1);                                                \
        bf16x8 pl0, pl1;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) { pl0[j] = (bf16)(S[j] - (float)pa0[j]); pl1[j] = (bf16)(S[8 + j] - (float)pa1[j]); } \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        _Pragma("unroll") for (int hs = 0; hs < 10; ++hs) {                                               \
            bf16x4* T_ = ft[hs % 3];                                                                      \
            bf16x8 v0, v1;                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { v0[j] = T_[0][j]; v0[4 + j] = T_[1][j]; v1[j] = T_[2][j]; v1[4 + j] = T_[3][j]; } \
            if ((hs & 1) == 0) {                                                                          \
                O[hs >> 1] = mfma_bf16(pl0, v0, O[hs >> 1]);                                              \
                O[hs >> 1] = mfma_bf16(pl1, v1, O[hs >> 1]);                                              \
                O[hs >> 1] = mfma_bf16(pa0, v0, O[hs >> 1]);                                              \
                O[hs >> 1] = mfma_bf16(pa1, v1, O[hs >> 1]);                                              \
            } else {                                                                                      \
                O[hs >> 1] = mfma_bf16(pa0, v0, O[hs >> 1]);                                              \
                O[hs >> 1] = mfma_bf16(pa1, v1, O[hs >> 1]);                                              \
            }                                                                                             \
            if (hs + 3 < 10) R3_LOADT(ft[hs % 3], (hs + 3) >> 1, (hs + 3) & 1);                           \
            __builtin_amdgcn_sched_barrier(0);                                                            \
        }                                                                                                 \
        bcur = bcur == 2 ? 0 : bcur + 1;                                                                  \
        ++i;                                                                                              \
    }
    int i = 0;
    while (i < nb_full) {
        R3_ITER(0)
        if (i >= nb_full) break;                           // workgroup-uniform
        R3_ITER(1)
    }
    if (nb_blocks > nb_full) {                             // the partial block (its table rows are already on their way)
        if (i & 1) { R3_TSLOW(1, blk_begin + i); R3_ITER(1) } else { R3_TSLOW(0, blk_begin + i); R3_ITER(0) }
    }
#undef R3_TSLOW
#undef R3_ITER
#undef R3_LOADT
#undef R3_TLOAD
#undef R3_STORE
#undef R3_LOAD
    float* o2 = a.pO2 + ((size_t)range * Bk + bk0) * HP;
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) o2[(size_t)acc_row(j, hh) * HP + 32 * nb + r32] = O[nb][j];
}

// teacher readout launch: x.ranges2 item ranges x (Bp - kd_row0) / 128 chunks.  false: the shape is not k_lx3r's (caller falls back)
bool lx3r_supports(const Lx3Args& x) {
    return x.H == 150 && x.Np >= F3_FB && (x.ldt & 3) == 0 && (((uintptr_t)x.teacher) & 15) == 0 && (long)(x.vrows + 5 * F3_FB) * x.H * 4 < (1l << 31);      // (block offsets are 32-bit; the stream runs 4 blocks ahead)
}
int lx3r_launch(const Lx3Args& x, void* stream) {
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
    if (!f) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lx3r, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * X3B_IMG_B);
        if (e != hipSuccess) return (int)e;
        f = true;
    }
    hipLaunchKernelGGL(k_lx3r, dim3(x.ranges2 * ((x.Bp - x.kd_row0) / G3_ROWS)), dim3(256), 3 * X3B_IMG_B, (hipStream_t)stream, x);
    return 0;
}

bool lx3f_supports(int H) { return (H & 1) == 0 && H >= 8 && H <= HP && ((H & 7) == 0 || (H & 7) == 4 || (H & 7) == 6); }

// x.ranges must be ader_lbf_ranges(x.N, x.Bp); Bp % 128 == 0
int lx3g_launch(const Lx3Args& x, void* stream) {
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
    if (!f) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lx3g<150>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * X3B_IMG_B);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)k_lx3g<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * X3B_IMG_B);
        if (e != hipSuccess) return (int)e;
        f = true;
    }
    if ((long)x.vrows * x.H * 4 >= (1l << 31)) return -2;          // the block offsets of the buffer loads are 32-bit
    if (x.H == 150) hipLaunchKernelGGL(k_lx3g<150>, dim3(x.ranges * (x.Bp / G3_ROWS)), dim3(256), 3 * X3B_IMG_B, (hipStream_t)stream, x);
    else hipLaunchKernelGGL(k_lx3g<0>, dim3(x.ranges * (x.Bp / G3_ROWS)), dim3(256), 3 * X3B_IMG_B, (hipStream_t)stream, x);
    return 0;
}

int lx3p_launch(const Lx3Args& x, void* stream) {
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
    if (!f) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lx3p<150>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * X3B_IMG_B);
        if (e != hipSuccess) return (int)e;
        f = true;
    }
    if ((long)x.vrows * x.H * 4 >= (1l << 31)) return -2;
    if (x.H != 150) return lx3g_launch(x, stream);                 // (the pipelined form exists for the reference's hidden size only)
    hipLaunchKernelGGL(k_lx3p<150>, dim3(x.ranges * (x.Bp / G3_ROWS)), dim3(256), 3 * X3B_IMG_B, (hipStream_t)stream, x);
    return 0;
}
