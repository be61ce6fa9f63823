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
//   (table_update.hip) dE tile = dlogit^T . rep per item-tile workgroup: S = rep.E^T recomputed (batch rows on the MFMA
//              rows), p = exp2(S*log2e + off_b) packed to bf16 in registers and fed back as the A operand of
//              dE[item,:] += P^T . rep with rep read k-major by ds_read_b64_tr_b16.  Each dE row is written once.
//   k_lbf_target_fix  the sparse one-hot term: dE[label_b] -= w_b * rep_b.
//
// Measured and dropped (round 2, cfg-S, 0.36 ms for the kept form): (a) a software-pipelined form that issues block i+1's S MFMAs
// before block i's softmax (0.40 ms: the second S accumulator costs the occupancy that hid the exp2 chain); (b) the same forward on
// v_mfma_f32_16x16x32_bf16 with 8 waves x 16 batch rows (126 VGPRs, 16 waves per CU; bit-equal results): 0.43 ms -- every MFMA of
// that shape fetches the same LDS operand bytes for half the flops, and the LDS pipe, not occupancy, is the limit here.
// (c) 64 batch rows per wave at one wave per SIMD (every table fragment feeds two MFMAs: half the LDS bytes per flop): 0.79 ms --
// 160 accumulator registers of O plus the rep fragments exceed the 256 architectural VGPRs, the compiler parks O in AGPRs and
// moves it back and forth for the softmax / rescale VALU work (~1900 v_accvgpr moves per three blocks), with no second wave to hide it.
//
// LDS tiles are row-major bf16 with a 168-element (336 B) row stride: ds_read_b128 operand reads are conflict-free
// (20 r mod 64 covers 16 distinct 4-bank slots).  gfx950 only.
#include <stdlib.h>
#include "lbf_common.h"
#include "x3_image.h"
#include "../../include/ader_hip.h"

// rep fp32 [B,H] -> rep_bf [Bp, LDR] bf16, zero padded (rows >= B, cols >= H)
__global__ __launch_bounds__(256) void k_lbf_prep(const float* __restrict__ rep, bf16* __restrict__ rep_bf, int B, int Bp, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp * LDR) return;
    const int b = i / LDR, c = i - b * LDR;
    rep_bf[i] = (bf16)((b < B && c < H) ? rep[(size_t)b * H + c] : 0.0f);
}

// The same for the padded row layout of a distilled step: rows [0, n_train) then rows [kd_row0, kd_row0 + n_ex) of rep_bf come from
// the compact rep [n_train + n_ex, H]; everything else is zero.  Also fills the per-row info of that layout: label (0 for KD and
// padding rows), loss weight, teacher row (-1: none) and the teacher's log-sum-exp in the log2 domain.
__global__ __launch_bounds__(256) void k_lbf_prep_kd(const float* __restrict__ rep, bf16* __restrict__ rep_bf, bf16* __restrict__ rep_lo,
                                                     int n_train, int n_ex,
                                                     int kd_row0, int Bp, int H, const int* __restrict__ pos,
                                                     const int* __restrict__ ex_trow, const float* __restrict__ tlse_all, float w_train,
                                                     float w_ex, int* __restrict__ lab, float* __restrict__ wrow, int* __restrict__ trow,
                                                     float* __restrict__ tlse2, char* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp * LDR) return;
    const int b = i / LDR, c = i - b * LDR;
    int src = -1;
    if (b < n_train) src = b;
    else if (b >= kd_row0 && b - kd_row0 < n_ex) src = n_train + (b - kd_row0);
    {
        const float x = (src >= 0 && c < H) ? rep[(size_t)src * H + c] : 0.0f;
        const bf16 h = (bf16)x;
        rep_bf[i] = h;
        if (rep_lo) {                                          // x3 mode: low-order operand rows
            const bf16 l = (bf16)(x - (float)h);
            rep_lo[i] = l;
            if (img && c < HP) {                               // ... and the fused update's operand images (as k_lx3_prep)
                char* p = img + (size_t)(b >> 5) * X3_IMG_B + x3_kc_off(c >> 3) + 16 * (b & 31) + 2 * (c & 7);
                *(bf16*)p = h;
                *(bf16*)(p + X3_PLANE_B) = l;
            }
        }
    }
    if (c == 0) {
        int l = 0, tr = -1; float w = 0.0f, tl = 0.0f;
        if (b < n_train) { l = pos[b]; w = (l > 0) ? w_train : 0.0f; }
        else if (src >= 0) {
            tr = ex_trow[b - kd_row0];
            if (tr >= 0) { w = w_ex; tl = tlse_all[tr] * LOG2E; } else tr = -1;
        }
        lab[b] = l; wrow[b] = w; trow[b] = tr; tlse2[b] = tl;
    }
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
    // distilled (KD) rows, ADER.py:132-137: rows [kd_row0, Bp) (kd_row0 % 128 == 0; = Bp: none) are exemplar rows whose softmax
    // runs over the first Np items only and whose target is softmax(teacher row)
    int kd_row0, Np;
    int n_train, n_ex;          // valid rows: [0, n_train) and [kd_row0, kd_row0 + n_ex); compact row of b >= kd_row0: n_train + b - kd_row0
    const float* teacher; long ldt;     // teacher logits [*, ldt] fp32, row trow[b] for batch row b
    const int* trow;            // [Bp] teacher row of a KD row (-1: padding)
    const float* tlse2;         // [Bp] log2-domain log-sum-exp of the teacher row over [0, Np)
    float* pO2;                 // [ranges2][Bp - kd_row0][HP] teacher readout partials: sum_j softmax(t)_j * E_j
    int ranges2;                // item ranges of the readout (= ranges when it shares the forward's launch)
};

static inline void lbf_no_kd(LbfArgs& a) {
    a.kd_row0 = a.Bp; a.Np = 0; a.n_train = a.B; a.n_ex = 0; a.teacher = nullptr; a.ldt = 0; a.trow = nullptr; a.tlse2 = nullptr;
    a.pO2 = nullptr; a.ranges2 = 0;
}

#define FB 32                      // items per streamed block
#define PCS_ROW (LDR * 2 / 16)     // 16-byte pieces per shadow row (21)
#define PCS_BLK (FB * PCS_ROW)     // pieces per block (672)
#define PPT 3                      // pieces per thread per block (256*3 >= 672)
#define RD 3                       // register ring depth: table blocks in flight per workgroup

// Streaming of 32-item table blocks: global -> register ring (RD blocks in flight) -> double-buffered LDS tile.
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

// Teacher readout of 128 KD rows over one item range:  O2[b,:] = sum_j softmax(teacher_b)_j * E_j  (j < Np), accumulated on the
// matrix cores exactly like the softmax-weighted readout of the forward (probabilities as the A operand, the table block read
// k-major from LDS).  With it the distillation term needs no second softmax pass:  sum_j pt_j * s_j = rep . O2  and
// dRep = w (O1/l - O2)  (ADER.py:135-137).  Partials go to pO2 [range][kd row][HP].
__device__ __forceinline__ void lbf_teacher_readout(const LbfArgs& a, bf16* E_l, int range, int b0, int blk_begin, int nb_blocks,
                                                    int Np) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    f32x16 O[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    const int tr = a.trow[b0 + r];
    const float tl2 = a.tlse2[b0 + r];
    const float* trp = a.teacher + (size_t)(tr < 0 ? 0 : tr) * a.ldt;
    const bool vec = ((a.ldt & 3) == 0) && (((uintptr_t)a.teacher & 15) == 0);
    // teacher logits of this lane's batch row for its 16 items of a block (4 runs of 4 consecutive items)
#define LBF_TLOAD(t_, blk_)                                                                              \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                      \
        const int it = (blk_) * FB + 8 * g + 4 * hh;                                                     \
        if (vec && it + 3 < Np) {                                                                        \
            const float4 v = *(const float4*)(trp + it);                                                 \
            t_[4 * g] = v.x; t_[4 * g + 1] = v.y; t_[4 * g + 2] = v.z; t_[4 * g + 3] = v.w;              \
        } else {                                                                                         \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) t_[4 * g + k] = (it + k < Np) ? trp[it + k] : -INFINITY; \
        }                                                                                                \
    }
    uint4 ring[RD][PPT];
    float tc[16], tn[16];
#pragma unroll
    for (int s_ = 0; s_ < RD; ++s_) if (s_ < nb_blocks) LBF_LOAD(s_, blk_begin + s_);
    if (nb_blocks > 0) LBF_TLOAD(tc, blk_begin);
    int cur = 0, i = 0;
    while (i < nb_blocks) {
#pragma unroll
        for (int s_ = 0; s_ < RD; ++s_) {
            if (i >= nb_blocks) break;                  // workgroup-uniform
            const int blk = blk_begin + i;
            const int i0 = blk * FB;
            LBF_STORE(s_, cur);
            if (i + RD < nb_blocks) LBF_LOAD(s_, blk + RD);
            if (i + 1 < nb_blocks) { LBF_TLOAD(tn, blk + 1); }
            __syncthreads();
            const bf16* Eb = E_l + cur * FB * LDR;
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // reg j of lane (r, hh) = item i0 + acc_row(j, hh) = i0 + 8 (j >> 2) + 4 hh + (j & 3): tc[] is in that order
                const float x = (tr >= 0) ? __builtin_amdgcn_exp2f(fmaf(tc[j], LOG2E, -tl2)) : 0.0f;
                S[j] = (i0 + acc_row(j, hh) < Np) ? x : 0.0f;
            }
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
#pragma unroll
            for (int j = 0; j < 16; ++j) tc[j] = tn[j];
            cur ^= 1;
            ++i;
        }
    }
    float* o = a.pO2 + ((size_t)range * (a.Bp - a.kd_row0) + (b0 - a.kd_row0)) * HP;
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[(size_t)acc_row(j, hh) * HP + 32 * nb + r] = O[nb][j];
}

__global__ __launch_bounds__(256, 2) void k_lbf_fwd(LbfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* E_l = (bf16*)smem_raw;                       // [2][FB][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // chunks: Bp/128 softmax chunks, then one TEACHER-READOUT chunk per 128 KD rows (same item ranges, same XCD grouping, so
    // the table blocks they stream are shared through L2)
    // The readout chunks come after all softmax workgroups and have their OWN, finer partition of the item blocks (ranges2 >=
    // ranges): a readout block costs ~1.5 softmax blocks (teacher loads + exp2 per element), and with a shared partition the
    // readout workgroups were the stragglers of the launch (+128 exemplar rows at 900k columns: 0.85 ms -> see DESIGN.md).
    const int nsm = a.Bp >> 7, nkd = (a.Bp - a.kd_row0) >> 7;
    const int n_sm = a.ranges * nsm;                                  // softmax workgroups (a multiple of 8)
    const bool readout = (int)blockIdx.x >= n_sm;
    const int bid = readout ? blockIdx.x - n_sm : blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int nranges = readout ? a.ranges2 : a.ranges;
    const int range = xcd + 8 * (slot / (readout ? nkd : nsm));
    int bc = readout ? (a.kd_row0 >> 7) + slot % nkd : slot % nsm;
    if (range >= nranges) return;
    const int N = (bc * 128 >= a.kd_row0) ? a.Np : a.N;              // columns of this chunk's softmax
    const int nblk_all = (a.N + FB - 1) / FB;                         // ranges partition the blocks of the whole catalog
    const int per = (nblk_all + nranges - 1) / nranges;
    const int blk_begin = range * per, blk_end = min((N + FB - 1) / FB, blk_begin + per);
    const int nb_blocks = max(0, blk_end - blk_begin);
    const int b0 = bc * 128 + wave * 32;
    if (readout) { lbf_teacher_readout(a, E_l, range, b0, blk_begin, nb_blocks, N); return; }
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
// X3: the target logit and the target row come from the fp32 operands (emb1_f = table row of item 1, rep_f [B,H]).
template <bool X3>
__global__ __launch_bounds__(640) void k_lbf_combine(LbfArgs a, const int* __restrict__ lab, const float* __restrict__ wrow,
                                                     float* __restrict__ lse, float* __restrict__ off, float* __restrict__ rowloss,
                                                     float* __restrict__ drep, const float* __restrict__ emb1_f,
                                                     const float* __restrict__ rep_f, AderLnfBwd lnf) {
    __shared__ float sc[1024];          // per-range scale 2^(pm - M) (0 for empty ranges)
    __shared__ float red[640], red2[640];
    __shared__ float sM, sL;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int R = a.ranges, H = a.H;
    const bool kd = b >= a.kd_row0;
    if (kd ? (b - a.kd_row0 >= a.n_ex) : (b >= a.n_train)) {     // padding rows: no loss, no gradient
        if (tid == 0) { lse[b] = 0.0f; rowloss[b] = 0.0f; off[b] = -INFINITY; }
        return;
    }
    const int bc_ = kd ? a.n_train + (b - a.kd_row0) : b;        // row of the compact [n_train + n_ex, H] tensors (rep_f, drep)
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
    if (tid < 64) {                                  // fixed order: ten strided terms per lane, then the wave butterfly (a single thread
        float v = 0.0f;                              // walking the 640 partials was a quarter of this kernel on the shipped catalogs)
#pragma unroll
        for (int k = 0; k < 10; ++k) v += red[tid + 64 * k];
        v = wave_sum(v);
        if (tid == 0) sL = v;
    }
    __syncthreads();
    const float L = sL;
    const float lse2 = M + log2f(L);
    const int t = lab[b] - 1;
    const float w = wrow[b];
    const int g = tid / 160, h = tid - g * 160;
    float oh = 0.0f, o2 = 0.0f;
    if (h < H) {
#pragma unroll 8
        for (int i = g; i < R; i += 4) oh += a.pO[((size_t)i * a.Bp + b) * HP + h] * sc[i];
        if (kd) {
            // distilled row: the teacher readout O2 = sum_j softmax(t)_j E_j, range partials summed like the student's (four
            // interleaved groups, fixed order; a single thread walking all ranges was 0.16 ms of dependent loads at 512 ranges)
            const int Bk = a.Bp - a.kd_row0;
#pragma unroll 8
            for (int i = g; i < a.ranges2; i += 4) o2 += a.pO2[((size_t)i * Bk + (b - a.kd_row0)) * HP + h];
        }
    }
    __syncthreads();
    red[tid] = oh;
    red2[tid] = o2;
    __syncthreads();
    float part = 0.0f, et = 0.0f;
    if (tid < H) {
        oh = ((red[tid] + red[160 + tid]) + red[320 + tid]) + red[480 + tid];
        if (kd) {
            // the target "row" of a distilled row is O2, and  sum_j pt_j s_j = rep . O2  with the operands the MFMA path multiplied
            et = ((red2[tid] + red2[160 + tid]) + red2[320 + tid]) + red2[480 + tid];
            part = (X3 ? rep_f[(size_t)bc_ * H + tid] : (float)a.rep_bf[(size_t)b * LDR + tid]) * et;
        } else if (t >= 0) {                         // target logit with the same bf16-rounded operands as the MFMA path
            if (X3) { et = emb1_f[(size_t)t * H + tid]; part = rep_f[(size_t)bc_ * H + tid] * et; }
            else { et = (float)a.sh1[(size_t)t * LDR + tid]; part = (float)a.rep_bf[(size_t)b * LDR + tid] * et; }
        }
        drep[(size_t)bc_ * H + tid] = w * (oh / L - et);
    }
    if (lnf.x) {
        // ---- fused backward of the final LayerNorm (ADER.py:83-85 differentiated; k_ln_bwd of rowwise.hip for this row, the same
        //      operations in the same order -- lane c sums channels c, c + 64, c + 128, then the wave butterfly -- so dx is bit-equal
        //      to the separate launch): dx = (dxh - mean(dxh) - xh mean(dxh xh)) / sd, dxh = g gamma, xh = (x - mean) / sd, g = dRep.
        //      gamma / beta partials of the row go to slab[row][2][H] (reduced later, beside the table update).
        __syncthreads();
        const float g = (tid < H) ? w * (oh / L - et) : 0.0f;
        const float sd = lnf.std[bc_];
        const float xh = (tid < H) ? (lnf.x[(size_t)bc_ * H + tid] - lnf.mean[bc_]) / sd : 0.0f;
        const float dxh = (tid < H) ? g * lnf.gamma[tid] : 0.0f;
        red[tid] = dxh;
        red2[tid] = dxh * xh;
        __syncthreads();
        if (tid < 64) {
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) { s1 += red[tid + 64 * i]; s2 += red2[tid + 64 * i]; }
            s1 = wave_sum(s1) / (float)H;
            s2 = wave_sum(s2) / (float)H;
            if (tid == 0) { sM = s1; sL = s2; }
        }
        __syncthreads();
        if (tid < H) {
            lnf.dx[(size_t)bc_ * H + tid] = (dxh - sM - xh * sL) / sd;
            lnf.slab[((size_t)bc_ * 2 + 0) * H + tid] = g * xh;
            lnf.slab[((size_t)bc_ * 2 + 1) * H + tid] = g;
        }
    }
    __syncthreads();
    red[tid] = (tid < H) ? part : 0.0f;
    __syncthreads();
    if (tid < 64) {
        float s_lab = (red[tid] + red[tid + 64]) + red[tid + 128];      // (H <= 160: three terms per lane, fixed order)
        s_lab = wave_sum(s_lab);
        if (tid != 0) return;
        const float z = lse2 / LOG2E;
        lse[b] = z;
        rowloss[b] = (t >= 0 || kd) ? w * (z - s_lab) : 0.0f;
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
template <bool X3>      // X3: rep_f = fp32 representations [B,H] and the label row enters in fp32 (float32-grade target logit)
__global__ __launch_bounds__(256) void k_lbf_merge_parts(const float* __restrict__ parts, int W, int Bp, int B, int H,
                                                         const float* __restrict__ e_lab, const bf16* __restrict__ rep_bf,
                                                         const float* __restrict__ rep_f,
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
            const float et = X3 ? e_lab[(size_t)b * H + c] : (float)(bf16)e_lab[(size_t)b * H + c];
            dot += (X3 ? rep_f[(size_t)b * H + c] : (float)rep_bf[(size_t)b * LDR + c]) * et;
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


// ============================================================================================= x3: float32-grade logits
// The same flash forward with every product as three bf16 MFMAs (hi.hi + lo.hi + hi.lo, operands split into
// bf16(x) + bf16(x - bf16(x)): ~2^-16 relative per product, fp32 accumulate) -- the reference's float32 logits
// (ADER.py:91-93) on the bf16 matrix cores.  There is no bf16 shadow in this mode: the fp32 table rows are streamed
// directly (600 B per item instead of 2 x 336 B) and split on the way into LDS.
// img != NULL: also the LDS images of the 32-row chunks that the fused update streams (ader_x3_rep_image's output, x3_image.h)
__global__ __launch_bounds__(256) void k_lx3_prep(const float* __restrict__ rep, bf16* __restrict__ rep_hi, bf16* __restrict__ rep_lo,
                                                  int B, int Bp, int H, char* __restrict__ img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp * LDR) return;
    const int b = i / LDR, c = i - b * LDR;
    const float x = ((b < B && c < H) ? rep[(size_t)b * H + c] : 0.0f) * X3_SR;      // (X3_SR = 1 in the product build)
    const bf16 h = (bf16)x, l = (bf16)(x - (float)h);
    rep_hi[i] = h;
    rep_lo[i] = l;
    if (img && c < HP) {
        char* p = img + (size_t)(b >> 5) * X3_IMG_B + x3_kc_off(c >> 3) + 16 * (b & 31) + 2 * (c & 7);
        *(bf16*)p = h;
        *(bf16*)(p + X3_PLANE_B) = l;
    }
}


#define XPPT 5                     // 16-byte fp32 vectors per thread per 32-item block (5 * 1024 floats >= 32 * 160)
// XRD = register ring depth, OCC = workgroups per CU the register budget is sized for (256 / 512 registers per lane)
template <int XRD, int OCC, bool READOUT = false>
__global__ __launch_bounds__(256, OCC) void k_lx3_fwd(Lx3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* E_l = (bf16*)smem_raw;                       // [2 buffers][2 planes: hi, lo][FB][LDR]
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // READOUT = false: the Bp/128 softmax chunks; READOUT = true (own launch: its registers would otherwise push the softmax path
    // into spills): one teacher-readout chunk per 128 KD rows, O2 = sum_j softmax(teacher)_j E_j as in lbf_teacher_readout
    const int nchunk = READOUT ? (a.Bp - a.kd_row0) >> 7 : a.Bp >> 7;
    const int nranges = READOUT ? a.ranges2 : a.ranges;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range = xcd + 8 * (slot / nchunk);
    int bc = slot % nchunk;
    if (range >= nranges) return;
    constexpr bool readout = READOUT;
    if (readout) bc += a.kd_row0 >> 7;
    const int H = a.H;
    const int N = (bc * 128 >= a.kd_row0) ? a.Np : a.N;              // columns of this chunk's softmax
    const int nblk_all = ((READOUT ? N : a.N) + FB - 1) / FB;          // (the readout partitions only the Np columns it reads)
    const int per = (nblk_all + nranges - 1) / nranges;
    const int blk_begin = range * per, blk_end = min((N + FB - 1) / FB, blk_begin + per);
    const int nb_blocks = max(0, blk_end - blk_begin);
    const int b0 = bc * 128 + wave * 32;
    // K padding (columns >= H) of both planes of both buffers stays zero: the block stores never touch it
    for (int i = tid; i < 2 * 2 * FB * LDR / 8; i += 256) ((uint4*)E_l)[i] = make_uint4(0u, 0u, 0u, 0u);
    bf16x8 bh_[10], bl_[10];
    if (!readout) {
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            bh_[ks] = *(const bf16x8*)(a.rep_hi + (size_t)(b0 + r) * LDR + 16 * ks + 8 * hh);
            bl_[ks] = *(const bf16x8*)(a.rep_lo + (size_t)(b0 + r) * LDR + 16 * ks + 8 * hh);
        }
    }
    f32x16 O[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    // A block = FB consecutive table rows = one flat run of FB*H floats whose 16-byte phase is the table's (H even): vector j
    // of thread t covers floats e = head + 4 t + 1024 j; its LDS position (row, col) does not depend on the block
    const int ph = (int)(((uintptr_t)a.emb1 & 15) >> 2);
    const int head = ph ? 4 - ph : 0;
    const int nfl = FB * H;
    int lo0[XPPT], lo1[XPPT];                           // LDS element offsets of floats (e, e+1) and (e+2, e+3); -1: outside
#pragma unroll
    for (int j = 0; j < XPPT; ++j) {
        const int e = head + 4 * tid + 1024 * j;
        const int row = e / H, col = e - row * H;
        lo0[j] = (e < nfl) ? row * LDR + col : -1;
        const int c1 = col + 2;
        lo1[j] = (e + 2 < nfl) ? ((c1 >= H) ? (row + 1) * LDR + (c1 - H) : row * LDR + c1) : -1;
    }
    f32x4_t ring[XRD][XPPT];
    f32x2_t ringh[XRD];
#define LX3_LOAD(slot_, blk_)                                                                            \
    {                                                                                                    \
        const float* src_ = a.emb1 + (size_t)(blk_) * FB * H;                                            \
        const int nav_ = min(FB, a.vrows - (blk_) * FB) * H;                                             \
        _Pragma("unroll") for (int j = 0; j < XPPT; ++j) {                                               \
            const int e_ = head + 4 * tid + 1024 * j;                                                    \
            f32x4_t v_ = (f32x4_t){0.f, 0.f, 0.f, 0.f};                                                  \
            if (e_ + 3 < nav_) v_ = *(const f32x4_t*)(src_ + e_);                                        \
            else if (e_ + 1 < nav_) { const f32x2_t t_ = *(const f32x2_t*)(src_ + e_); v_[0] = t_[0]; v_[1] = t_[1]; } \
            ring[slot_][j] = v_;                                                                         \
        }                                                                                                \
        ringh[slot_] = (head && tid == 0 && nav_ > 0) ? *(const f32x2_t*)src_ : (f32x2_t){0.f, 0.f};     \
    }
#define LX3_PUT2(dst_, off_, x0_, x1_)                                                                   \
    {                                                                                                    \
        const bf16 h0_ = (bf16)(x0_), h1_ = (bf16)(x1_);                                                 \
        bf16x2 vh_, vl_;                                                                                 \
        vh_[0] = h0_; vh_[1] = h1_; vl_[0] = (bf16)((x0_) - (float)h0_); vl_[1] = (bf16)((x1_) - (float)h1_); \
        *(bf16x2*)((dst_) + (off_)) = vh_;                                                               \
        *(bf16x2*)((dst_) + FB * LDR + (off_)) = vl_;                                                    \
    }
#define LX3_STORE(slot_, buf_)                                                                           \
    {                                                                                                    \
        bf16* dst_ = E_l + (buf_) * 2 * FB * LDR;                                                        \
        _Pragma("unroll") for (int j = 0; j < XPPT; ++j) {                                               \
            if (lo0[j] >= 0) LX3_PUT2(dst_, lo0[j], ring[slot_][j][0], ring[slot_][j][1]);               \
            if (lo1[j] >= 0) LX3_PUT2(dst_, lo1[j], ring[slot_][j][2], ring[slot_][j][3]);               \
        }                                                                                                \
        if (head && tid == 0) LX3_PUT2(dst_, 0, ringh[slot_][0], ringh[slot_][1]);                       \
    }
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    if constexpr (READOUT) {
        // teacher readout of 128 KD rows (see lbf_teacher_readout): O2 += PT^T . E with PT = softmax(teacher) split hi/lo
        const int tr = a.trow[b0 + r];
        const float tl2 = a.tlse2[b0 + r];
        const float* trp = a.teacher + (size_t)(tr < 0 ? 0 : tr) * a.ldt;
        const bool vec = ((a.ldt & 3) == 0) && (((uintptr_t)a.teacher & 15) == 0);
        const int Np = N;
        float tc[16], tn[16];
#pragma unroll
        for (int s_ = 0; s_ < XRD; ++s_) if (s_ < nb_blocks) LX3_LOAD(s_, blk_begin + s_);
#if defined(LX3RO_KO) && LX3RO_KO == 1
#define RO_TLOAD(t_, blk_) { _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) t_[j_] = -3.0f; }
#else
#define RO_TLOAD(t_, blk_) LBF_TLOAD(t_, blk_)
#endif
        if (nb_blocks > 0) RO_TLOAD(tc, blk_begin);
        __syncthreads();
        int cur = 0, i = 0;
        while (i < nb_blocks) {
#pragma unroll
            for (int s_ = 0; s_ < XRD; ++s_) {
                if (i >= nb_blocks) break;
                const int blk = blk_begin + i;
                const int i0 = blk * FB;
                LX3_STORE(s_, cur);
#if !(defined(LX3RO_KO) && LX3RO_KO == 2)
                if (i + XRD < nb_blocks) LX3_LOAD(s_, blk + XRD);
#endif
                if (i + 1 < nb_blocks) { RO_TLOAD(tn, blk + 1); }
                __syncthreads();
                const bf16* Eh = E_l + cur * 2 * FB * LDR;
                f32x16 S;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float x = (tr >= 0) ? __builtin_amdgcn_exp2f(fmaf(tc[j], LOG2E, -tl2)) : 0.0f;
                    S[j] = (i0 + acc_row(j, hh) < Np) ? x : 0.0f;
                }
                const bf16x8 pa0 = pack8(S, 0), pa1 = pack8(S, 1);
                bf16x8 pl0, pl1;
#pragma unroll
                for (int j = 0; j < 8; ++j) { pl0[j] = (bf16)(S[j] - (float)pa0[j]); pl1[j] = (bf16)(S[8 + j] - (float)pa1[j]); }
#if defined(LX3RO_KO) && LX3RO_KO == 3
                if (a.H == 1)
#endif
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) {
                    const bf16* base = Eh + (4 * hh + q4) * LDR + 32 * nb + 16 * g1 + 4 * p4;
                    const bf16x4 l0 = tr_read(base), h0 = tr_read(base + 8 * LDR);
                    const bf16x4 l1 = tr_read(base + 16 * LDR), h1 = tr_read(base + 24 * LDR);
                    bf16x8 b0v, b1v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { b0v[j] = l0[j]; b0v[4 + j] = h0[j]; b1v[j] = l1[j]; b1v[4 + j] = h1[j]; }
                    O[nb] = mfma_bf16(pa0, b0v, O[nb]);
                    O[nb] = mfma_bf16(pa1, b1v, O[nb]);
                    O[nb] = mfma_bf16(pl0, b0v, O[nb]);
                    O[nb] = mfma_bf16(pl1, b1v, O[nb]);
                    const bf16* bl = base + FB * LDR;
                    const bf16x4 m0 = tr_read(bl), n0 = tr_read(bl + 8 * LDR);
                    const bf16x4 m1 = tr_read(bl + 16 * LDR), n1 = tr_read(bl + 24 * LDR);
                    bf16x8 c0v, c1v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { c0v[j] = m0[j]; c0v[4 + j] = n0[j]; c1v[j] = m1[j]; c1v[4 + j] = n1[j]; }
                    O[nb] = mfma_bf16(pa0, c0v, O[nb]);
                    O[nb] = mfma_bf16(pa1, c1v, O[nb]);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) tc[j] = tn[j];
                cur ^= 1;
                ++i;
            }
        }
        float* o2 = a.pO2 + ((size_t)range * (a.Bp - a.kd_row0) + (b0 - a.kd_row0)) * HP;
#pragma unroll
        for (int nb = 0; nb < 5; ++nb)
#pragma unroll
            for (int j = 0; j < 16; ++j) o2[(size_t)acc_row(j, hh) * HP + 32 * nb + r] = O[nb][j];
        return;
    }
    else {
#pragma unroll
    for (int s_ = 0; s_ < XRD; ++s_) if (s_ < nb_blocks) LX3_LOAD(s_, blk_begin + s_);
    __syncthreads();                                    // zero fill done before the first block store
    int cur = 0, i = 0;
    while (i < nb_blocks) {
#pragma unroll
        for (int s_ = 0; s_ < XRD; ++s_) {
            if (i >= nb_blocks) break;                  // workgroup-uniform
            const int blk = blk_begin + i;
            LX3_STORE(s_, cur);
            if (i + XRD < nb_blocks) LX3_LOAD(s_, blk + XRD);
            __syncthreads();
            const bf16* Eh = E_l + cur * 2 * FB * LDR;
            const bf16* El = Eh + FB * LDR;
            const int i0 = blk * FB;
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) S[j] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                const bf16x8 ah = *(const bf16x8*)(Eh + r * LDR + 16 * ks + 8 * hh);
                const bf16x8 al = *(const bf16x8*)(El + r * LDR + 16 * ks + 8 * hh);
                S = mfma_bf16(ah, bh_[ks], S);
                S = mfma_bf16(al, bh_[ks], S);
                S = mfma_bf16(ah, bl_[ks], S);
            }
            if (i0 + FB > N) {                          // tail block: items >= N are outside the softmax
#pragma unroll
                for (int j = 0; j < 16; ++j) if (i0 + acc_row(j, hh) >= N) S[j] = -INFINITY;
            }
            float tmax = S[0];
#pragma unroll
            for (int j = 1; j < 16; ++j) tmax = fmaxf(tmax, S[j]);
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
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const bf16* base = Eh + (4 * hh + q4) * LDR + 32 * nb + 16 * g1 + 4 * p4;
                const bf16x4 l0 = tr_read(base), h0 = tr_read(base + 8 * LDR);
                const bf16x4 l1 = tr_read(base + 16 * LDR), h1 = tr_read(base + 24 * LDR);
                bf16x8 b0v, b1v;
#pragma unroll
                for (int j = 0; j < 4; ++j) { b0v[j] = l0[j]; b0v[4 + j] = h0[j]; b1v[j] = l1[j]; b1v[4 + j] = h1[j]; }
                O[nb] = mfma_bf16(pa0, b0v, O[nb]);
                O[nb] = mfma_bf16(pa1, b1v, O[nb]);
                O[nb] = mfma_bf16(pl0, b0v, O[nb]);
                O[nb] = mfma_bf16(pl1, b1v, O[nb]);
                const bf16* bl = base + FB * LDR;
                const bf16x4 m0 = tr_read(bl), n0 = tr_read(bl + 8 * LDR);
                const bf16x4 m1 = tr_read(bl + 16 * LDR), n1 = tr_read(bl + 24 * LDR);
                bf16x8 c0v, c1v;
#pragma unroll
                for (int j = 0; j < 4; ++j) { c0v[j] = m0[j]; c0v[4 + j] = n0[j]; c1v[j] = m1[j]; c1v[4 + j] = n1[j]; }
                O[nb] = mfma_bf16(pa0, c0v, O[nb]);
                O[nb] = mfma_bf16(pa1, c1v, O[nb]);
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
}

// the x3 forwards of logits_x3.hip on conflict-free block images: k_lx3g (32x32x16, any supported H), k_lx3p (the same with the
// vector work inside the MFMA phases; H = 150: the default) and the teacher readout of distilled steps k_lx3r
bool lx3f_supports(int H);
int lx3g_launch(const Lx3Args& x, void* stream);
int lx3p_launch(const Lx3Args& x, void* stream);
// lx3r_supports: H = 150, 16-byte aligned teacher rows, at least one whole block
bool lx3r_supports(const Lx3Args& x);
int lx3r_launch(const Lx3Args& x, void* stream);
// ADER_X3_FWD = old | g | p: the round-2 kernel (k_lx3_fwd), k_lx3g, or k_lx3p (default; falls back to k_lx3g for H != 150)
// (read only by diagnostic builds, -DADER_DIAG: a production build takes no kernel choice from the environment)
static int lx3_env() {
#ifdef ADER_DIAG
    static int v = -1;
    if (v < 0) { const char* e = getenv("ADER_X3_FWD"); v = !e ? 4 : (e[0] == 'o' ? 0 : (e[0] == 'g' ? 2 : 4)); }
    return v;
#else
    return 4;
#endif
}
// 2: the block-image kernels (Bp % 128 == 0 and a supported H), 0: k_lx3_fwd
static int lx3_kind(int H, int Bp) {
    if (!lx3f_supports(H)) return 0;
    return (lx3_env() >= 2 && Bp % 128 == 0) ? 2 : 0;
}
static int lx3gh_launch(const Lx3Args& x, void* stream) {
    return lx3_env() == 4 ? lx3p_launch(x, stream) : lx3g_launch(x, stream);
}

// ============================================================================================= C ABI
static const size_t kFwdLds = (size_t)2 * FB * LDR * sizeof(bf16);

extern "C" {

int ader_lbf_ranges_kd(int N, int Bp, int kd_row0);
int ader_lbf_readout_ranges(int N, int Bp, int kd_row0);

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
    // item ranges per 128-row chunk: a multiple of 8 (blocks b and b + 8 share an XCD: the chunks of a range are placed on one),
    // as many as keep ranges * chunks within the 512 resident workgroups (2 per CU), at least one 32-item block each
    const int nblk = (N + FB - 1) / FB;
    const int nchunk = Bp / 128;
    int r = (512 / (nchunk < 1 ? 1 : nchunk)) / 8 * 8;
    if (r > (nblk + 7) / 8 * 8) r = (nblk + 7) / 8 * 8;
    if (r < 8) r = 8;
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
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
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
    a.pm = pm; a.pl = pl; a.pO = pO; a.off = nullptr; a.demb1 = nullptr; lbf_no_kd(a);
    hipLaunchKernelGGL(k_lbf_prep, dim3((Bp * LDR + 255) / 256), dim3(256), 0, st, rep, (bf16*)rep_bf, B, Bp, H);
    hipLaunchKernelGGL(k_lbf_fwd, dim3(a.ranges * (Bp / 128)), dim3(256), kFwdLds, st, a);
    hipLaunchKernelGGL(k_lbf_combine<false>, dim3(Bp), dim3(640), 0, st, a, lab, wrow, lse, off, rowloss, drep, (const float*)nullptr,
                       (const float*)nullptr, AderLnfBwd{});
    hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, st, rowloss, B, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Forward of a DISTILLED step (ADER.py:108-137) on the bf16 flash path: n_train one-hot rows (labels pos, weight w_train) and n_ex
// exemplar rows distilled against softmax(teacher[ex_trow[e], :Np]) (weight w_ex = lambda / n_ex), student softmax over the first Np
// items only.  rep is the compact [n_train + n_ex, H] tensor (exemplar rows last, main.py:229); inside, rows are laid out
// [train rows padded to 128 | exemplar rows padded to 128] (Bp = both paddings; kd_row0 = first exemplar row): lab / wrow / trow /
// tlse2 / lse / off / rowloss are [Bp] in THAT layout (they feed ader_tab_update_sh_kd), drep is compact.  tlse_all[r] = natural
// log-sum-exp of teacher row r over [0, Np).  Scratch as ader_lbf_fwd plus pO2: ranges * (Bp - kd_row0) * 160 floats.
int ader_lbf_fwd_kd(const float* rep, const void* shadow, int item_num, int n_train, int n_ex, int kd_row0, int Bp, int H, int N,
                    int Np, const int* pos, const int* ex_trow, const float* teacher, long ldt, const float* tlse_all, float w_train,
                    float w_ex, int* lab, float* wrow, int* trow, float* tlse2, void* rep_bf, float* pm, float* pl, float* pO,
                    float* pO2, float* lse, float* off, float* rowloss, float* loss, float* drep, void* stream) {
    if (n_train + n_ex <= 0) return 0;
    if (Bp % 128 != 0 || kd_row0 % 128 != 0 || n_train > kd_row0 || kd_row0 + n_ex > Bp || H > HP || (H & 1) || H < 2 || N > item_num ||
        Np > N || Np < 1) return -2;
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
    if (!f) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lbf_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds);
        if (e != hipSuccess) return (int)e;
        f = true;
    }
    hipStream_t st = (hipStream_t)stream;
    LbfArgs a;
    a.sh1 = (const bf16*)shadow + LDR; a.vrows = item_num; a.tile_off = 0;
    a.rep_bf = (const bf16*)rep_bf; a.B = n_train + n_ex; a.Bp = Bp; a.H = H; a.N = N;
    const int nsm = Bp / 128, nkd = (Bp - kd_row0) / 128;
    a.ranges = ader_lbf_ranges_kd(N, Bp, kd_row0);
    a.pm = pm; a.pl = pl; a.pO = pO; a.off = nullptr; a.demb1 = nullptr;
    a.kd_row0 = kd_row0; a.Np = Np; a.n_train = n_train; a.n_ex = n_ex; a.teacher = teacher; a.ldt = ldt; a.trow = trow; a.tlse2 = tlse2;
    a.pO2 = pO2; a.ranges2 = ader_lbf_readout_ranges(N, Bp, kd_row0);
    hipLaunchKernelGGL(k_lbf_prep_kd, dim3((Bp * LDR + 255) / 256), dim3(256), 0, st, rep, (bf16*)rep_bf, (bf16*)nullptr, n_train, n_ex,
                       kd_row0, Bp, H, pos, ex_trow, tlse_all, w_train, w_ex, lab, wrow, trow, tlse2, (char*)nullptr);
    hipLaunchKernelGGL(k_lbf_fwd, dim3(a.ranges * nsm + a.ranges2 * nkd), dim3(256), kFwdLds, st, a);
    hipLaunchKernelGGL(k_lbf_combine<false>, dim3(Bp), dim3(640), 0, st, a, (const int*)lab, (const float*)wrow, lse, off, rowloss, drep,
                       (const float*)nullptr, (const float*)nullptr, AderLnfBwd{});
    hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, st, rowloss, Bp, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}
// ranges used by ader_lbf_fwd_kd (scratch sizing): pm / pl: R*Bp, pO: R*Bp*160 with R = ader_lbf_ranges_kd; pO2: R2*(Bp-kd_row0)*160
// with R2 = ader_lbf_readout_ranges (the readout chunks take the workgroup slots the softmax chunks leave of the 512 resident ones)
int ader_lbf_ranges_kd(int N, int Bp, int kd_row0) { return ader_lbf_ranges(N, (Bp / 128 + (Bp - kd_row0) / 128) * 128); }
int ader_lbf_readout_ranges(int N, int Bp, int kd_row0) {
    const int nsm = Bp / 128, nkd = (Bp - kd_row0) / 128, R = ader_lbf_ranges_kd(N, Bp, kd_row0);
    if (nkd < 1) return R;
    const int nblk = (N + FB - 1) / FB;
    int extra = ((512 - (nsm + nkd) * R) / nkd) / 8 * 8;
    if (extra < 0) extra = 0;
    int r2 = R + extra;
    if (r2 > (nblk + 7) / 8 * 8) r2 = (nblk + 7) / 8 * 8;
    return r2 < R ? R : r2;
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
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
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
    a.pm = pm; a.pl = pl; a.pO = pO; a.off = nullptr; a.demb1 = nullptr; lbf_no_kd(a);
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
    hipLaunchKernelGGL(k_lbf_merge_parts<false>, dim3((Bp + 3) / 4), dim3(256), 0, st, parts, world, Bp, B, H, e_lab,
                       (const bf16*)rep_bf, (const float*)nullptr, wrow, lse, off, rowloss, drep);
    hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, st, rowloss, B, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}

// x3 operand rows of the representations: rep_hi = bf16(rep), rep_lo = bf16(rep - rep_hi), [Bp,168] each, zero padded
int ader_lx3_prep(const float* rep, void* rep_hi, void* rep_lo, int B, int Bp, int H, void* stream) {
    if (Bp <= 0) return 0;
    if (B > Bp || H > HP) return -2;
    hipLaunchKernelGGL(k_lx3_prep, dim3((Bp * LDR + 255) / 256), dim3(256), 0, (hipStream_t)stream, rep, (bf16*)rep_hi, (bf16*)rep_lo,
                       B, Bp, H, (char*)nullptr);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Forward of the one-hot softmax CE over items 1..N at float32 grade (three bf16 MFMAs per product), streaming the fp32
// table itself.  Same scratch and outputs as ader_lbf_fwd; rep_hi / rep_lo: Bp*168 bf16 each.
static int lx3_attr() {
    static bool f_dev[ADER_MAX_DEV] = {};
    bool& f = f_dev[ader_cur_dev()];
    if (!f) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lx3_fwd<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((size_t)2 * 2 * FB * LDR * sizeof(bf16)));
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)k_lx3_fwd<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)2 * 2 * FB * LDR * sizeof(bf16)));
        if (e != hipSuccess) return (int)e;
        f = true;
    }
    return 0;
}

int ader_lx3_fwd_img(const float* rep, const float* emb, int item_num, int B, int Bp, int H, int N, const int* lab, const float* wrow,
                     void* rep_hi, void* rep_lo, float* pm, float* pl, float* pO, float* lse, float* off, float* rowloss, float* loss,
                     float* drep, void* rep_img, void* stream);
int ader_lx3_fwd(const float* rep, const float* emb, int item_num, int B, int Bp, int H, int N, const int* lab, const float* wrow,
                 void* rep_hi, void* rep_lo, float* pm, float* pl, float* pO, float* lse, float* off, float* rowloss, float* loss,
                 float* drep, void* stream) {
    return ader_lx3_fwd_img(rep, emb, item_num, B, Bp, H, N, lab, wrow, rep_hi, rep_lo, pm, pl, pO, lse, off, rowloss, loss, drep, nullptr,
                            stream);
}
// ... rep_img != NULL: ader_x3_rep_image's output (ader_x3_rep_image_bytes(Bp) bytes, 16-byte aligned, pads zero) is written by the same
// launch that cuts the operand planes: the fused update finds it ready (one launch and one kernel boundary fewer on the critical path)
int ader_lx3_fwd_img(const float* rep, const float* emb, int item_num, int B, int Bp, int H, int N, const int* lab, const float* wrow,
                     void* rep_hi, void* rep_lo, float* pm, float* pl, float* pO, float* lse, float* off, float* rowloss, float* loss,
                     float* drep, void* rep_img, void* stream) {
    return ader_lx3_fwd_img_lnf(rep, emb, item_num, B, Bp, H, N, lab, wrow, rep_hi, rep_lo, pm, pl, pO, lse, off, rowloss, loss, drep, rep_img,
                                nullptr, stream);
}
// ... with the backward of the FINAL LayerNorm fused into the merge (lnf != NULL; include/ader_hip.h AderLnfBwd): rep = LN_f(x) (ADER.py:83-85),
// so the row that has just formed dRep also forms dx = LN_f'(dRep) -- one launch and one kernel boundary fewer on the critical path of every step
int ader_lx3_fwd_img_lnf(const float* rep, const float* emb, int item_num, int B, int Bp, int H, int N, const int* lab, const float* wrow,
                         void* rep_hi, void* rep_lo, float* pm, float* pl, float* pO, float* lse, float* off, float* rowloss, float* loss,
                         float* drep, void* rep_img, const AderLnfBwd* lnf, void* stream) {
    if (B <= 0) return 0;
    if (rep_img && ((uintptr_t)rep_img & 15)) return -2;
    if (Bp % 128 != 0 || B > Bp || H > HP || (H & 1) || H < 2 || N > item_num || ((uintptr_t)emb & 7)) return -2;
    const size_t lds = (size_t)2 * 2 * FB * LDR * sizeof(bf16);
    int rc = lx3_attr();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    Lx3Args x;
    x.emb1 = emb + H; x.vrows = item_num; x.rep_hi = (const bf16*)rep_hi; x.rep_lo = (const bf16*)rep_lo;
    const int nk = lx3_kind(H, Bp);
    x.Bp = Bp; x.H = H; x.N = N; x.ranges = ader_lbf_ranges(N, Bp); x.pm = pm; x.pl = pl; x.pO = pO;
    x.kd_row0 = Bp; x.Np = 0; x.teacher = nullptr; x.ldt = 0; x.trow = nullptr; x.tlse2 = nullptr; x.pO2 = nullptr; x.ranges2 = 0;
    LbfArgs a;
    a.sh1 = nullptr; a.vrows = item_num; a.tile_off = 0; a.rep_bf = (const bf16*)rep_hi; a.B = B; a.Bp = Bp; a.H = H; a.N = N;
    a.ranges = x.ranges; a.pm = pm; a.pl = pl; a.pO = pO; a.off = nullptr; a.demb1 = nullptr; lbf_no_kd(a);
    hipLaunchKernelGGL(k_lx3_prep, dim3((Bp * LDR + 255) / 256), dim3(256), 0, st, rep, (bf16*)rep_hi, (bf16*)rep_lo, B, Bp, H, (char*)rep_img);
    if (nk) { rc = lx3gh_launch(x, stream); if (rc) return rc; }
    else hipLaunchKernelGGL((k_lx3_fwd<2, 2>), dim3(x.ranges * (Bp / 128)), dim3(256), lds, st, x);
    hipLaunchKernelGGL(k_lbf_combine<true>, dim3(Bp), dim3(640), 0, st, a, lab, wrow, lse, off, rowloss, drep, emb + H, rep,
                       lnf ? *lnf : AderLnfBwd{});
    if (loss) hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, st, rowloss, B, loss);     // NULL: ader_lbf_sum later (off the critical path)
    HIP_LAUNCH_CHECK();
    return 0;
}

// loss = sum of the per-row losses a flash forward left in rowloss[0..n) (fixed order), for callers that passed loss = NULL
int ader_lbf_sum(const float* rowloss, int n, float* loss, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, rowloss, n, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}

// x3 distilled forward: scratch sizes.  pm / pl / pO use R = ader_lbf_ranges(N, Bp); the teacher readout is a launch of its own with
// R2 = ader_lx3_readout_ranges(Np, Bp - kd_row0) item ranges: pO2 holds R2 * (Bp - kd_row0) * 160 floats.
int ader_lx3_readout_ranges(int Np, int Bk) {
    // ... but at least four 32-item blocks per range: on the shipped catalogs (Np ~ 25-43 k) 512 ranges of one or two blocks made the
    // merge (k_lbf_combine) sum 512 partials per exemplar row for a readout of a few microseconds
    int r = ader_lbf_ranges(Np, Bk);
    const int cap = (((Np + FB - 1) / FB + 3) / 4 + 7) / 8 * 8;
    if (r > cap) r = cap;
    return r < 8 ? 8 : r;
}

// ader_lbf_fwd_kd at float32 grade: the distilled step's forward on the x3 kernels (arguments as ader_lbf_fwd_kd, with the fp32
// table instead of the shadow and the two operand planes rep_hi / rep_lo).
int ader_lx3_fwd_kd(const float* rep, const float* emb, int item_num, int n_train, int n_ex, int kd_row0, int Bp, int H, int N,
                    int Np, const int* pos, const int* ex_trow, const float* teacher, long ldt, const float* tlse_all, float w_train,
                    float w_ex, int* lab, float* wrow, int* trow, float* tlse2, void* rep_hi, void* rep_lo, float* pm, float* pl,
                    float* pO, float* pO2, float* lse, float* off, float* rowloss, float* loss, float* drep, void* stream) {
    return ader_lx3_fwd_kd_lnf(rep, emb, item_num, n_train, n_ex, kd_row0, Bp, H, N, Np, pos, ex_trow, teacher, ldt, tlse_all, w_train, w_ex,
                               lab, wrow, trow, tlse2, rep_hi, rep_lo, pm, pl, pO, pO2, lse, off, rowloss, loss, drep, nullptr, nullptr, stream);
}
int ader_lx3_fwd_kd_lnf(const float* rep, const float* emb, int item_num, int n_train, int n_ex, int kd_row0, int Bp, int H, int N,
                        int Np, const int* pos, const int* ex_trow, const float* teacher, long ldt, const float* tlse_all, float w_train,
                        float w_ex, int* lab, float* wrow, int* trow, float* tlse2, void* rep_hi, void* rep_lo, float* pm, float* pl,
                        float* pO, float* pO2, float* lse, float* off, float* rowloss, float* loss, float* drep, void* rep_img,
                        const AderLnfBwd* lnf, void* stream) {
    if (n_train + n_ex <= 0) return 0;
    if (Bp % 128 != 0 || kd_row0 % 128 != 0 || n_train > kd_row0 || kd_row0 + n_ex > Bp || H > HP || (H & 1) || H < 2 || N > item_num ||
        Np > N || Np < 1 || ((uintptr_t)emb & 7) || ((uintptr_t)rep_img & 15)) return -2;
    const size_t lds = (size_t)2 * 2 * FB * LDR * sizeof(bf16);
    int rc = lx3_attr();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    Lx3Args x;
    x.emb1 = emb + H; x.vrows = item_num; x.rep_hi = (const bf16*)rep_hi; x.rep_lo = (const bf16*)rep_lo;
    const int nk = lx3_kind(H, Bp);
    x.Bp = Bp; x.H = H; x.N = N; x.ranges = ader_lbf_ranges(N, Bp); x.pm = pm; x.pl = pl; x.pO = pO;
    x.kd_row0 = kd_row0; x.Np = Np; x.teacher = teacher; x.ldt = ldt; x.trow = trow; x.tlse2 = tlse2; x.pO2 = pO2;
    x.ranges2 = ader_lx3_readout_ranges(Np, Bp - kd_row0);
    LbfArgs a;
    a.sh1 = nullptr; a.vrows = item_num; a.tile_off = 0; a.rep_bf = (const bf16*)rep_hi; a.B = n_train + n_ex; a.Bp = Bp; a.H = H; a.N = N;
    a.ranges = x.ranges; a.pm = pm; a.pl = pl; a.pO = pO; a.off = nullptr; a.demb1 = nullptr;
    a.kd_row0 = kd_row0; a.Np = Np; a.n_train = n_train; a.n_ex = n_ex; a.teacher = teacher; a.ldt = ldt; a.trow = trow; a.tlse2 = tlse2;
    a.pO2 = pO2; a.ranges2 = x.ranges2;
    hipLaunchKernelGGL(k_lbf_prep_kd, dim3((Bp * LDR + 255) / 256), dim3(256), 0, st, rep, (bf16*)rep_hi, (bf16*)rep_lo, n_train, n_ex,
                       kd_row0, Bp, H, pos, ex_trow, tlse_all, w_train, w_ex, lab, wrow, trow, tlse2, (char*)rep_img);
    if (nk) { rc = lx3gh_launch(x, stream); if (rc) return rc; }
    else hipLaunchKernelGGL((k_lx3_fwd<2, 2>), dim3(x.ranges * (Bp / 128)), dim3(256), lds, st, x);
    if (lx3r_supports(x)) { rc = lx3r_launch(x, stream); if (rc) return rc; }
    else hipLaunchKernelGGL((k_lx3_fwd<2, 2, true>), dim3(x.ranges2 * ((Bp - kd_row0) / 128)), dim3(256), lds, st, x);
    hipLaunchKernelGGL(k_lbf_combine<true>, dim3(Bp), dim3(640), 0, st, a, (const int*)lab, (const float*)wrow, lse, off, rowloss, drep,
                       emb + H, rep, lnf ? *lnf : AderLnfBwd{});
    if (loss) hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, st, rowloss, Bp, loss);     // NULL: ader_lbf_sum(rowloss, Bp) later
    HIP_LAUNCH_CHECK();
    return 0;
}

// ---- catalog-sharded data parallelism at float32 grade (x3): the counterparts of ader_lbf_fwd_shard / ader_lbf_merge_parts.
// Softmax partials of ALL Bp batch rows (operand planes rep_hi / rep_lo [Bp,168] of the all-gathered representations,
// ader_lx3_prep) over the item shard [item_begin+1, item_begin+item_count] (clipped to N), streaming the fp32 table rows of the
// shard.  part [Bp][152] = {M (log2 domain), L, O[0..H)}; scratch pm/pl/pO sized with ader_lbf_ranges(item_count, Bp).
int ader_lx3_fwd_shard(const void* rep_hi, const void* rep_lo, const float* emb, int item_num, int Bp, int H, int N, int item_begin,
                       int item_count, float* pm, float* pl, float* pO, float* part, void* stream) {
    if (Bp <= 0) return 0;
    if (Bp % 128 != 0 || H > HP || (H & 1) || H < 2 || N > item_num || item_begin < 0 || ((uintptr_t)emb & 7)) return -2;
    hipStream_t st = (hipStream_t)stream;
    int n_loc = N - item_begin;
    if (n_loc > item_count) n_loc = item_count;
    if (n_loc < 0) n_loc = 0;
    Lx3Args x;
    x.emb1 = emb + (size_t)H * (1 + item_begin); x.vrows = item_num - item_begin;
    x.rep_hi = (const bf16*)rep_hi; x.rep_lo = (const bf16*)rep_lo;
    x.Bp = Bp; x.H = H; x.N = n_loc; x.ranges = n_loc > 0 ? ader_lbf_ranges(n_loc, Bp) : 0; x.pm = pm; x.pl = pl; x.pO = pO;
    x.kd_row0 = Bp; x.Np = 0; x.teacher = nullptr; x.ldt = 0; x.trow = nullptr; x.tlse2 = nullptr; x.pO2 = nullptr; x.ranges2 = 0;
    LbfArgs a;
    a.sh1 = nullptr; a.vrows = x.vrows; a.tile_off = 0; a.rep_bf = (const bf16*)rep_hi; a.B = Bp; a.Bp = Bp; a.H = H; a.N = n_loc;
    a.ranges = x.ranges; a.pm = pm; a.pl = pl; a.pO = pO; a.off = nullptr; a.demb1 = nullptr; lbf_no_kd(a);
    if (x.ranges > 0) {
        int rc = 0;
        if (lx3_kind(H, Bp) == 2) rc = lx3gh_launch(x, stream);
        else {
            rc = lx3_attr();
            if (!rc) hipLaunchKernelGGL((k_lx3_fwd<2, 2>), dim3(x.ranges * (Bp / 128)), dim3(256), (size_t)2 * 2 * FB * LDR * sizeof(bf16), st, x);
        }
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_lbf_combine_partial, dim3(Bp), dim3(640), 0, st, a, part);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Distilled rows under the catalog-sharded scheme (ADER.py:132-137 with the table sharded over the ranks).  The student's softmax
// partials of the exemplar rows over the rank's items below Np come from ader_lx3_fwd_shard called with N = Np; the TEACHER readout
// O2 = sum_j softmax(teacher)_j E_j splits over the item shards as plain partial sums (every term is normalised by the teacher's
// full log-sum-exp): this launcher adds up the rank's items [item_begin + 1, item_begin + item_count] (clipped to Np) for the Bk
// gathered exemplar rows -- part2 [Bk][152], channels at [2, 2 + H) like the student partials; scratch pO2 sized
// ader_lx3_readout_ranges(item_count, Bk) * Bk * 160 floats.  trow [Bk] teacher row (-1: padding row), tlse2 [Bk] its log2-domain
// log-sum-exp over [0, Np).
__global__ __launch_bounds__(256) void k_lx3_sum_ranges(const float* __restrict__ pO2, int ranges, int Bk, int H, float* __restrict__ part2) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= PART_LD) return;
    float v = 0.0f;
    if (c >= 2 && c - 2 < H) {
#pragma unroll 8
        for (int i = 0; i < ranges; ++i) v += pO2[((size_t)i * Bk + b) * HP + (c - 2)];
    }
    part2[(size_t)b * PART_LD + c] = v;
}
int ader_lx3_readout_shard(const float* emb, int item_num, int Bk, int H, int Np, int item_begin, int item_count, const float* teacher,
                           long ldt, const int* trow, const float* tlse2, float* pO2, float* part2, void* stream) {
    if (Bk <= 0) return 0;
    if (Bk % 128 != 0 || H > HP || (H & 1) || H < 2 || Np > item_num || item_begin < 0 || (item_begin & 3) || ((uintptr_t)emb & 7) ||
        !teacher || !trow || !tlse2) return -2;
    hipStream_t st = (hipStream_t)stream;
    int n_loc = Np - item_begin;
    if (n_loc > item_count) n_loc = item_count;
    if (n_loc < 0) n_loc = 0;
    int ranges2 = 0;
    if (n_loc > 0) {
        int rc = lx3_attr();
        if (rc) return rc;
        Lx3Args x;
        x.emb1 = emb + (size_t)H * (1 + item_begin); x.vrows = item_num - item_begin;
        x.rep_hi = nullptr; x.rep_lo = nullptr;
        x.Bp = Bk; x.H = H; x.N = n_loc; x.ranges = 0; x.pm = nullptr; x.pl = nullptr; x.pO = nullptr;
        x.kd_row0 = 0; x.Np = n_loc; x.teacher = teacher + item_begin; x.ldt = ldt; x.trow = trow; x.tlse2 = tlse2; x.pO2 = pO2;
        ranges2 = ader_lx3_readout_ranges(n_loc, Bk);
        x.ranges2 = ranges2;
        if (lx3r_supports(x)) { rc = lx3r_launch(x, stream); if (rc) return rc; }
        else hipLaunchKernelGGL((k_lx3_fwd<2, 2, true>), dim3(x.ranges2 * (Bk / 128)), dim3(256), (size_t)2 * 2 * FB * LDR * sizeof(bf16), st, x);
    }
    hipLaunchKernelGGL(k_lx3_sum_ranges, dim3(Bk), dim3(256), 0, st, (const float*)pO2, ranges2, Bk, H, part2);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Cross-rank merge for the distilled rows of THIS rank: parts_s / parts_t [W][Bk][152] = the W ranks' student partials (over the
// first Np items) and teacher-readout partial sums -> lse, backward offset, loss row w (lse - rep . O2) and dRep = w (O1 / l - O2)
// (k_lbf_combine's distilled branch with the range partials replaced by rank partials).  rep [B,H] fp32 representations of the
// rank's exemplar rows, wrow [Bk] (0 for padding rows).  One wave per row; fixed summation order.
__global__ __launch_bounds__(256) void k_lx3_merge_parts_kd(const float* __restrict__ parts_s, const float* __restrict__ parts_t, int W,
                                                            int Bk, int B, int H, const float* __restrict__ rep,
                                                            const float* __restrict__ wrow, float* __restrict__ lse,
                                                            float* __restrict__ off, float* __restrict__ rowloss,
                                                            float* __restrict__ drep) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= Bk) return;
    const float w = b < B ? wrow[b] : 0.0f;
    if (b >= B) {
        if (lane == 0) { lse[b] = 0.0f; rowloss[b] = 0.0f; off[b] = -INFINITY; }
        return;
    }
    float M = -INFINITY;
    for (int i = 0; i < W; ++i) M = fmaxf(M, parts_s[((size_t)i * Bk + b) * PART_LD]);
    float L = 0.0f, o[3] = {0.0f, 0.0f, 0.0f}, o2[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < W; ++i) {
        const float* pr = parts_s + ((size_t)i * Bk + b) * PART_LD;
        const float* pt = parts_t + ((size_t)i * Bk + b) * PART_LD;
        const float m = pr[0];
        const float sc = (m != -INFINITY) ? __builtin_amdgcn_exp2f(m - M) : 0.0f;
        L += pr[1] * sc;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int c = lane + 64 * k; if (c < H) { o[k] += pr[2 + c] * sc; o2[k] += pt[2 + c]; } }
    }
    const float lse2 = M + log2f(L);
    float dot = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int c = lane + 64 * k;
        if (c < H) {
            dot += rep[(size_t)b * H + c] * o2[k];
            drep[(size_t)b * H + c] = w * (o[k] / L - o2[k]);
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
int ader_lx3_merge_parts_kd(const float* parts_s, const float* parts_t, int world, int Bk, int B, int H, const float* rep,
                            const float* wrow, float* lse, float* off, float* rowloss, float* drep, void* stream) {
    if (Bk <= 0) return 0;
    if (B > Bk || H > 192 || world < 1) return -2;
    hipLaunchKernelGGL(k_lx3_merge_parts_kd, dim3((Bk + 3) / 4), dim3(256), 0, (hipStream_t)stream, parts_s, parts_t, world, Bk, B, H, rep,
                       wrow, lse, off, rowloss, drep);
    HIP_LAUNCH_CHECK();
    return 0;
}

// Merge of the W ranks' partials as ader_lbf_merge_parts, with the target logit in fp32: rep [B,H] fp32 representations of THIS
// rank's rows, e_lab [B,H] fp32 table rows of their labels.
int ader_lx3_merge_parts(const float* parts, int world, int Bp, int B, int H, const float* e_lab, const float* rep,
                         const float* wrow, float* lse, float* off, float* rowloss, float* loss, float* drep, void* stream) {
    if (Bp <= 0) return 0;
    if (B > Bp || H > 192 || world < 1) return -2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_lbf_merge_parts<true>, dim3((Bp + 3) / 4), dim3(256), 0, st, parts, world, Bp, B, H, e_lab,
                       (const bf16*)nullptr, rep, wrow, lse, off, rowloss, drep);
    hipLaunchKernelGGL(k_lbf_sum, dim3(1), dim3(256), 0, st, rowloss, B, loss);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
