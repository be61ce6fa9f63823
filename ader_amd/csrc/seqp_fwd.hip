// The SASRec forward stack on PACKED session tiles: k_seq_fwd (seq_fwd.hip) for batches of short sessions.
//
// The reference pads every session to maxlen positions (util.py:161-169) and runs the whole stack on the padding (ADER.py:41-91);
// a padded position influences no real one (its key is masked, modules.py:188-193; its outputs are re-zeroed, ADER.py:80; its
// gradient is zero), so this kernel computes the REAL positions only.  ader_seq_pack_plan (seqp_plan.hip) lays them out in
// 64-row tiles -- several short sessions per tile, each a contiguous run of rows in position order -- and one workgroup owns a
// tile exactly as k_seq_fwd owns a session: same phases, same LDS images, same bf16x3 arithmetic, same weights streamed once per
// TILE instead of once per session.  What changes is bookkeeping:
//   * a row's identity (item id, position t, session b, dropout counter) comes from the plan's per-row records;
//   * attention is block-diagonal: query q sees the keys [first row of its session, q] (causal inside the session,
//     modules.py:196-202); keys of other sessions get probability exactly 0, as the masked keys of the reference do;
//   * activations are stored in tile order ([tile*64 + row, H]; the probabilities as [tile][key][query]); the backward kernels
//     (seqp_bwd.hip) read the same layout.  A pruned last block (only the last position of a session feeds the representation,
//     ADER.py:85) computes every row of the tile and STORES the last rows into the compact [B,..] tensors of the unpacked path;
//   * rep[b] is the final LayerNorm of the last row of session b.
// Identical results to k_seq_fwd up to the summation order of the softmax denominator (keys sit at other tile rows).
// heads == 1, T <= 64, H even and <= 150.  gfx950 only.
#include "seqp_common.h"

#ifdef SFP_STAMP     // diagnostic build only (tools/build_variant.sh ... -DSFP_STAMP): clocks per phase of waves 0 and 4 of each workgroup
__device__ unsigned long long sfp_dbg[2 * 40 * 1024];
#define SFS_INIT unsigned long long seg[40], tprev; for (int i_ = 0; i_ < 40; ++i_) seg[i_] = 0; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#define SFS(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                  __builtin_amdgcn_sched_barrier(0); seg[k_] += t_ - tprev; tprev = t_; }
#define SFB(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                  __builtin_amdgcn_sched_barrier(0); if (l == 0) seg[1 + k_] += t_ - tprev; else seg[17 + k_] += t_ - tprev; tprev = t_; }
#define SFS_DUMP { if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 1024) for (int k_ = 0; k_ < 40; ++k_) \
                       sfp_dbg[(blockIdx.x * 2 + (wave == 4)) * 40 + k_] = seg[k_]; }
extern "C" int ader_dbg_read_sfp(void* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sfp_dbg), (size_t)n * 8); }
#else
#define SFS_INIT
#define SFS(k_)
#define SFB(k_)
#define SFS_DUMP {}
#endif

struct SeqpCtx {
    bf16 *R0, *R1, *R2;
    float *km_l, *qm_l, *red_l;
    int *sq_l, *info_l, *tp_l;
    uint32_t *gph_l, *gpt_l;
    int nrows, nrb;
    size_t prow0;
#ifdef SFP_STAMP
    unsigned long long* seg;
    unsigned long long* tprev;
#endif
};

// The blocks of the stack on one tile.  SMALL: the 16-column mapping of tiles with at most 32 rows; otherwise the 32x32 mapping of k_seq_fwd.
template <bool SMALL>
__device__ __forceinline__ void seqp_blocks(const AderSeqFwd& a, const SeqpCtx& cx) {
    bf16* const R0 = cx.R0; bf16* const R1 = cx.R1; bf16* const R2 = cx.R2;
    float* const Xf = (float*)R2;
    float* const km_l = cx.km_l; float* const qm_l = cx.qm_l; float* const red_l = cx.red_l;
    int* const sq_l = cx.sq_l; int* const info_l = cx.info_l; int* const tp_l = cx.tp_l;
    uint32_t* const gph_l = cx.gph_l; uint32_t* const gpt_l = cx.gpt_l;
    const int nrows = cx.nrows, nrb = cx.nrb;
    const size_t prow0 = cx.prow0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = wave % 5, mh = wave / 5;
    const int T = a.T, H = a.H;
    const uint32_t H4 = (uint32_t)H * 4u;
    const bool skipw = mh == 1 && nrows <= 32;        // this wave's 32 rows hold no position (32x32 mapping)
    const int npass = nrows > 40 ? 2 : 1;             // row-layout phases: rows 40 pass + 4 wave + rsub
    (void)tp_l; (void)gpt_l; (void)red_l; (void)nrb; (void)skipw; (void)T;
#ifdef SFP_STAMP
    unsigned long long* seg = cx.seg;
    unsigned long long& tprev = *cx.tprev;
#endif
    typedef const AderSeqBlock __attribute__((address_space(4))) * BlkPtr;
    const BlkPtr blks = (BlkPtr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
                                 offsetof(AderSeqFwd, blk));
    bf16x8 bh[10], bl[10];
    if (SMALL) load_bfrags16((const bf16*)blks[0].w[0], wave, lane, bh, bl);                  // Wq of block 0: in flight across LayerNorm 1
    else load_bfrags((const bf16*)blks[0].w[0], nb, lane & 31, lane >> 5, bh, bl);
#pragma unroll 1
    for (int l = 0; l < a.L; ++l) {
        const BlkPtr kp = blks + l;
#define k (*kp)
        const bool pruned = k.pruned != 0;      // last block: only a session's last position keeps its query / FFN path (ADER.py:85)
        const int mrows = pruned ? a.B : nrows; // rows of the per-row tensors of the query / FFN path (compact or tile)
        const size_t mrow0 = pruned ? 0 : prow0;
        float g1[10], be1[10], bias5[5];
        load10(k.ln1_g, H, lane & 15, g1); load10(k.ln1_b, H, lane & 15, be1);
        {
            const int n = SMALL ? 16 * wave + (lane & 15) : 32 * nb + (lane & 31);
#pragma unroll
            for (int i = 0; i < 5; ++i) bias5[i] = (n < H) ? k.bias[i][n] : 0.0f;
        }
        // ---- LN1 (ADER.py:67 via modules.py:44-48) + key/query masks (modules.py:188,208); x -> R0, LN(x) -> R1
        {
            const Out oq = make_rows(k.q_in, mrow0, mrows, H);
            const Out om = make_rows(k.mean1, mrow0, mrows, 1), os = make_rows(k.std1, mrow0, mrows, 1);
            const Out okm = make_rows(k.kmask, prow0, nrows, 1), oqm = make_rows(k.qmask, mrow0, mrows, 1);
            const int lane_q = opaque(lane);
            const int sub = lane_q & 15, rsub = lane_q >> 4;
#pragma unroll 1
            for (int pass = 0; pass < npass; ++pass) {
                const int t = 40 * pass + 4 * wave + rsub;
                if (t < nrows) {
                    float x[10], s = 0.0f;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int c = sub + 16 * i;
                        x[i] = (c < H) ? Xf[t * XS + c] : 0.0f;
                        s += x[i];
                    }
                    s = row16_sum(s);
                    const float mean = s / (float)H;
                    float q = 0.0f;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const float dlt = (sub + 16 * i < H) ? (x[i] - mean) : 0.0f;
                        q += dlt * dlt;
                    }
                    q = row16_sum(q);
                    const float sd = sqrtf(q / (float)H + LN_EPS);
                    const float rsd = 1.0f / sd;
                    float ys = 0.0f;
                    bf16* T0 = R0 + t * LDR + sub;
                    bf16* T1 = R1 + t * LDR + sub;
                    const uint32_t rb = row_base(pruned, t, H4, info_l);
                    const uint32_t bo = rb + (uint32_t)sub * 4u;
#pragma unroll
                    for (int i = 0; i < 10; ++i) put_split(T0, T0 + TR * LDR, 16 * i, x[i]);
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int c = sub + 16 * i;
                        const float y = (c < H) ? g1[i] * ((x[i] - mean) * rsd) + be1[i] : 0.0f;
                        ys += y;
                        put_split(T1, T1 + TR * LDR, 16 * i, y);
                        bstore(oq, (c < H) ? bo + 64u * i : OOB, y);
                    }
                    ys = row16_sum(ys);
                    const float kmv = (s != 0.0f) ? 1.0f : 0.0f, qmv = (ys != 0.0f) ? 1.0f : 0.0f;
                    if (sub == 0) { km_l[t] = kmv; qm_l[t] = qmv; }
                    const uint32_t so = (sub == 0) ? row_base(pruned, t, 4u, info_l) : OOB;
                    bstore(okm, (sub == 0) ? (uint32_t)t * 4u : OOB, kmv);
                    bstore(oqm, so, qmv); bstore(om, so, mean); bstore(os, so, sd);
                } else if (t < TR) {
                    if (sub == 0) { km_l[t] = 0.0f; qm_l[t] = 0.0f; }
                }
            }
            if (npass == 1 && tid < TR - 40) { km_l[40 + tid] = 0.0f; qm_l[40 + tid] = 0.0f; }
        }
        SFB(0)
        lds_barrier();
        SFB(1)
        float g2[10], be2[10];                  // LayerNorm 2 parameters: requested after the V phase, consumed after the attention
        if (SMALL) {
            // ======== tiles of at most 32 rows: ten waves x 16 output columns on v_mfma_f32_16x16x32_bf16 (header comment) ========
            const int lane_s = opaque(lane);
            const int c = lane_s & 15, g = lane_s >> 4;
            const int n = 16 * wave + c;
            const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
            // ---- Q = LN(x).Wq + bq -> memory, hi/lo -> R1 (in place)
            {
                f32x4v acc[2];
                tile_mma16(R1, c, g, bh, bl, nrb, acc);
                load_bfrags16((const bf16*)k.w[1], wave, lane_s, bh, bl);
                const Out o = make_rows(k.Q, mrow0, mrows, H);
                bf16* Th = R1 + (4 * g) * LDR + n;
                lds_barrier();                                          // every wave has read its R1 rows
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    if (rb >= nrb) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = (n < H) ? acc[rb][i] + bias5[0] : 0.0f;
                        put_split(Th, Th + TR * LDR, (16 * rb + i) * LDR, v);
                        bstore(o, row_base(pruned, 16 * rb + 4 * g + i, H4, info_l) + n4, v);
                    }
                }
            }
            // ---- K = x.Wk + bk -> memory, hi/lo -> R2 (the fp32 tile is dead)
            {
                f32x4v acc[2];
                tile_mma16(R0, c, g, bh, bl, nrb, acc);
                load_bfrags16((const bf16*)k.w[2], wave, lane_s, bh, bl);
                const Out o = make_rows(k.K, prow0, nrows, H);
                bf16* Th = R2 + (4 * g) * LDR + n;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    if (rb >= nrb) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = (n < H) ? acc[rb][i] + bias5[1] : 0.0f;
                        put_split(Th, Th + TR * LDR, (16 * rb + i) * LDR, v);
                        bstore(o, (uint32_t)(16 * rb + 4 * g + i) * H4 + n4, v);
                    }
                }
            }
            // ---- V = x.Wv + bv -> memory, hi/lo -> R0 (in place)
            {
                f32x4v acc[2];
                tile_mma16(R0, c, g, bh, bl, nrb, acc);
                const Out o = make_rows(k.V, prow0, nrows, H);
                bf16* Th = R0 + (4 * g) * LDR + n;
                lds_barrier();                                          // every wave has read its R0 rows
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    if (rb >= nrb) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = 16 * rb + 4 * g + i;
                        const float v = (n < H && t < nrows) ? acc[rb][i] + bias5[2] : 0.0f;   // unused rows: exact zeros (0 * V below)
                        put_split(Th, Th + TR * LDR, (16 * rb + i) * LDR, v);
                        bstore(o, (uint32_t)t * H4 + n4, v);
                    }
                }
            }
            __syncthreads();        // full barrier: LN(x) rows written to memory by other waves are re-read after the attention
            load10(k.ln2_g, H, lane & 15, g2); load10(k.ln2_b, H, lane & 15, be2);
            float qres[8];
            {
                const Out oq = make_rows(k.q_in, mrow0, mrows, H);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) qres[4 * rb + i] = bload(oq, row_base(pruned, 16 * rb + 4 * g + i, H4, info_l) + n4);
            }
            // ---- scores + softmax: S^T in 16x16 blocks -- wave 0: keys 0..15 x queries 0..15, wave 1: keys 16.. x queries 16..,
            //      wave 2: keys 0..15 x queries 16..; the block above the diagonal (wave 3) is all zeros
            bf16* Ph = R1;                                   // [queries][LDP] hi, then lo: overlays the Q tile once S is done
            bf16* Pl = R1 + TR * LDP;
            {
                const int kb2 = (wave == 1 || wave == 3) ? 1 : 0, qb2 = (wave == 1 || wave == 2) ? 1 : 0;
                const bool sact = wave < 3 && (wave == 0 || nrb == 2);
                const int q = 16 * qb2 + c, key0 = 16 * kb2 + 4 * g;
                const int seg0 = info_l[q] & 63;
                f32x4v S = {0.0f, 0.0f, 0.0f, 0.0f};
                float mx = -INFINITY, sum = 0.0f;
                if (sact) {
                    const bf16* Kh = R2 + (16 * kb2 + c) * LDR + 8 * g;
                    const bf16* Qh = R1 + (16 * qb2 + c) * LDR + 8 * g;
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) {
                        const bf16x8 ah = *(const bf16x8*)(Kh + 32 * ks), al = *(const bf16x8*)(Kh + TR * LDR + 32 * ks);
                        const bf16x8 qh = *(const bf16x8*)(Qh + 32 * ks), ql = *(const bf16x8*)(Qh + TR * LDR + 32 * ks);
                        S = mfma16_bf16(al, qh, S);
                        S = mfma16_bf16(ah, ql, S);
                        S = mfma16_bf16(ah, qh, S);
                    }
                    const float r_sqrt_dh = 1.0f / a.sqrt_dh;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = key0 + i;
                        float sc = S[i] * r_sqrt_dh;                             // modules.py:185
                        if (km_l[key] == 0.0f) sc = NEG_PAD;                     // modules.py:188-193
                        if (key >= seg0 && key <= q) mx = fmaxf(mx, sc);         // modules.py:196-202 inside the session
                        S[i] = sc;
                    }
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                }
                if (wave < 4 && g == 0) red_l[kb2 * 32 + q] = mx;
                lds_barrier();                               // also: every read of the Q and K tiles is done
                if (sact) {
                    mx = fmaxf(mx, red_l[(kb2 ^ 1) * 32 + q]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = key0 + i;
                        const float e = (key >= seg0 && key <= q) ? expf(S[i] - mx) : 0.0f;
                        S[i] = e;
                        sum += e;
                    }
                    sum += __shfl_xor(sum, 16, 64);
                    sum += __shfl_xor(sum, 32, 64);
                }
                if (wave < 4 && g == 0) red_l[64 + kb2 * 32 + q] = sum;
                lds_barrier();
                if (wave == 3) {                             // keys 16.. of queries 0..15: above the diagonal
                    const bf16x4 z4 = {(bf16)0.0f, (bf16)0.0f, (bf16)0.0f, (bf16)0.0f};
                    *(bf16x4*)(Ph + q * LDP + key0) = z4;
                    *(bf16x4*)(Pl + q * LDP + key0) = z4;
                } else if (sact) {
                    sum += red_l[64 + (kb2 ^ 1) * 32 + q];
                    const float r_sum = 1.0f / sum;
                    const float qm = qm_l[q];                                     // modules.py:208-211
                    const uint32_t athr = k.d_attn.thr, akey = k.d_attn.key;
                    const float ascale = athr ? k.d_attn.scale : 1.0f;
                    const Out op = pruned ? make_rows(k.P, 0, a.B, T) : make_rows(k.P, prow0, TR, TR);
                    const int qinf = info_l[q];
                    const uint32_t pq = pruned ? ((qinf & 64) ? (uint32_t)(qinf >> 16) * (uint32_t)T * 4u : OOBH)
                                               : ((q < nrows) ? (uint32_t)q * 4u : OOBH);
                    const uint32_t dq = gpt_l[q];
                    bf16x4 h4, l4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = key0 + i;
                        const bool in = key >= seg0 && key <= q;
                        float p = S[i] * r_sum;
                        const uint32_t tk = (uint32_t)tp_l[key];
                        const uint32_t pk_ = in ? (pruned ? tk * 4u : (uint32_t)key * (uint32_t)(TR * 4)) : OOBH;
                        bstore(op, pq + pk_, p);
                        p *= qm;
                        if (athr) p = ((lowbias32((dq + tk) ^ akey) >> 8) >= athr) ? p * ascale : 0.0f;     // modules.py:214
                        h4[i] = (bf16)p;
                        l4[i] = (bf16)(p - (float)h4[i]);
                    }
                    *(bf16x4*)(Ph + q * LDP + key0) = h4;
                    *(bf16x4*)(Pl + q * LDP + key0) = l4;
                }
            }
            lds_barrier();
            // ---- O = P_drop . V, x1 = O + LN(x) (modules.py:223); R2 is Xf again
            {
                f32x4v O[2];
                const int q4 = c >> 2, p4 = c & 3;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    O[rb] = (f32x4v){0.0f, 0.0f, 0.0f, 0.0f};
                    if (rb >= nrb) continue;
                    const bf16x8 ph = *(const bf16x8*)(Ph + (16 * rb + c) * LDP + 8 * g);
                    const bf16x8 pl = *(const bf16x8*)(Pl + (16 * rb + c) * LDP + 8 * g);
                    const bf16* Vp = R0 + (8 * g + q4) * LDR + 16 * wave + 4 * p4;
                    const bf16x8 vh = cat4(tr_read(Vp), tr_read(Vp + 4 * LDR));
                    const bf16x8 vl = cat4(tr_read(Vp + TR * LDR), tr_read(Vp + TR * LDR + 4 * LDR));
                    O[rb] = mfma16_bf16(pl, vh, O[rb]);
                    O[rb] = mfma16_bf16(ph, vl, O[rb]);
                    O[rb] = mfma16_bf16(ph, vh, O[rb]);
                }
                load_bfrags16((const bf16*)k.w[3], wave, lane_s, bh, bl);       // W1, consumed after LN2
                const Out o = make_rows(k.x1, mrow0, mrows, H);
                float* Xp = Xf + (4 * g) * XS + n;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    if (rb >= nrb) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = O[rb][i] + qres[4 * rb + i];
                        Xp[(16 * rb + i) * XS] = v;
                        bstore(o, row_base(pruned, 16 * rb + 4 * g + i, H4, info_l) + n4, v);
                    }
                }
            }
        } else {
        // ---- Q = LN(x).Wq + bq (modules.py:172) -> memory, hi/lo -> R1 (in place)
        {
            PHASE_IDS;
            f32x16 acc = tile_mma(R1, mh, r, hh, bh, bl, skipw);
            load_bfrags((const bf16*)k.w[1], nb, r, hh, bh, bl);
            SFB(2)
            const Out o = make_rows(k.Q, mrow0, mrows, H);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
            bf16* Th = R1 + t0 * LDR + n;
            lds_barrier();                                              // every wave has read its R1 rows
            if (!skipw) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float v = (n < H) ? acc[j] + bias5[0] : 0.0f;
                    put_split(Th, Th + TR * LDR, ROWJ(j) * LDR, v);
                    bstore(o, row_base(pruned, t0 + ROWJ(j), H4, info_l) + n4, v);
                }
            }
        }
        SFB(3)
        // ---- K = x.Wk + bk (modules.py:173) -> memory, hi/lo -> R2 (the fp32 tile is dead)
        {
            PHASE_IDS;
            f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
            load_bfrags((const bf16*)k.w[2], nb, r, hh, bh, bl);
            SFB(4)
            const Out o = make_rows(k.K, prow0, nrows, H);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
            bf16* Th = R2 + t0 * LDR + n;
            if (!skipw) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float v = (n < H) ? acc[j] + bias5[1] : 0.0f;
                    put_split(Th, Th + TR * LDR, ROWJ(j) * LDR, v);
                    bstore(o, boff0 + ROWJ(j) * H4, v);
                }
            }
        }
        SFB(5)
        // ---- V = x.Wv + bv (modules.py:174) -> memory, hi/lo -> R0 (in place)
        {
            PHASE_IDS;
            f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
            SFB(6)
            const Out o = make_rows(k.V, prow0, nrows, H);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
            bf16* Th = R0 + t0 * LDR + n;
            lds_barrier();                                              // every wave has read its R0 rows
            if (!skipw) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float v = (n < H && t0 + ROWJ(j) < nrows) ? acc[j] + bias5[2] : 0.0f;   // unused rows: exact zeros (0 * V below)
                    put_split(Th, Th + TR * LDR, ROWJ(j) * LDR, v);
                    bstore(o, boff0 + ROWJ(j) * H4, v);
                }
            }
        }
        SFB(7)
        __syncthreads();        // full barrier: LN(x) rows written to memory by other waves are re-read after the attention
        SFB(8)
        // ---- attention (modules.py:177-223), block-diagonal over the sessions of the tile.  Wave (mq, kb) of the first four owns
        //      the 32x32 block S^T[keys 32kb..][queries 32mq..]; block (0, 1) is above the diagonal: nothing to do.
        float qres[16];
        load10(k.ln2_g, H, lane & 15, g2); load10(k.ln2_b, H, lane & 15, be2);
        {
            PHASE_IDS;
            const Out oq = make_rows(k.q_in, mrow0, mrows, H);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
#pragma unroll
            for (int j = 0; j < 16; ++j) qres[j] = bload(oq, row_base(pruned, t0 + ROWJ(j), H4, info_l) + n4);    // residual rows, added after P.V
        }
        SFB(9)
        bf16* Ph = R1;                                   // [64 queries][LDP] hi, then lo: overlays the Q tile once S is done
        bf16* Pl = R1 + TR * LDP;
        {
            PHASE_IDS;
            const int mq = wave >> 1, kb = wave & 1;
            const bool swave = wave < 4 && !(mq == 1 && nrows <= 32);     // (queries 32.. unused: nobody reads their P rows)
            const bool sdiag = swave && !(mq == 0 && kb == 1);             // blocks that hold keys <= queries
            const int q = 32 * mq + r;
            const int key0 = 32 * kb + 4 * hh;           // key of register j: key0 + ROWJ(j)
            const int seg0 = swave ? (info_l[q] & 63) : 0;                  // first row of the query's session
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) S[j] = 0.0f;
            float mx = -INFINITY, sum = 0.0f;
            if (sdiag) {
                const bf16* Qh = R1 + q * LDR + 8 * hh;
                const bf16* Kh = R2 + (32 * kb + r) * LDR + 8 * hh;
#pragma unroll
                for (int ks = 0; ks < 10; ++ks) {
                    const bf16x8 qh = *(const bf16x8*)(Qh + 16 * ks), ql = *(const bf16x8*)(Qh + TR * LDR + 16 * ks);
                    const bf16x8 ah = *(const bf16x8*)(Kh + 16 * ks), al = *(const bf16x8*)(Kh + TR * LDR + 16 * ks);
                    S = mfma_bf16(al, qh, S);
                    S = mfma_bf16(ah, ql, S);
                    S = mfma_bf16(ah, qh, S);
                }
                const float r_sqrt_dh = 1.0f / a.sqrt_dh;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int key = key0 + ROWJ(j);
                    float sc = S[j] * r_sqrt_dh;                                 // modules.py:185
                    if (km_l[key] == 0.0f) sc = NEG_PAD;                         // modules.py:188-193
                    if (key >= seg0 && key <= q) mx = fmaxf(mx, sc);             // modules.py:196-202 inside the session
                    S[j] = sc;
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            }
            if (swave && hh == 0) red_l[kb * TR + q] = mx;
            lds_barrier();                               // also: every read of the Q and K tiles is done
            if (swave) {
                mx = fmaxf(mx, red_l[(kb ^ 1) * TR + q]);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int key = key0 + ROWJ(j);
                    const float e = (sdiag && key >= seg0 && key <= q) ? expf(S[j] - mx) : 0.0f;
                    S[j] = e;
                    sum += e;
                }
                sum += __shfl_xor(sum, 32, 64);
                if (hh == 0) red_l[2 * TR + kb * TR + q] = sum;
            }
            lds_barrier();
            if (sdiag) {                                 // (probabilities outside the query's session are exact zeros: never stored, the backward knows)
                sum += red_l[2 * TR + (kb ^ 1) * TR + q];
                const float r_sum = 1.0f / sum;
                const float qm = qm_l[q];                                         // modules.py:208-211
                const uint32_t athr = k.d_attn.thr, akey = k.d_attn.key;
                const float ascale = athr ? k.d_attn.scale : 1.0f;
                // P^T: [tile][key][query]; pruned: row b of [B][T], the probabilities of the session's last query over its positions
                const Out op = pruned ? make_rows(k.P, 0, a.B, T) : make_rows(k.P, prow0, TR, TR);
                const int qinf = info_l[q];
                const uint32_t pq = pruned ? ((qinf & 64) ? (uint32_t)(qinf >> 16) * (uint32_t)T * 4u : OOBH)
                                           : ((q < nrows) ? (uint32_t)q * 4u : OOBH);
                const uint32_t dq = gpt_l[q];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    bf16x4 h4, l4;
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        const int j = 4 * jj + j2;
                        const int key = key0 + ROWJ(j);
                        const bool in = key >= seg0 && key <= q;
                        float p = S[j] * r_sum;
                        const uint32_t tk = (uint32_t)tp_l[key];
                        const uint32_t pk_ = in ? (pruned ? tk * 4u : (uint32_t)key * (uint32_t)(TR * 4)) : OOBH;
                        bstore(op, pq + pk_, p);
                        p *= qm;
                        if (athr) p = ((lowbias32((dq + tk) ^ akey) >> 8) >= athr) ? p * ascale : 0.0f;     // modules.py:214
                        h4[j2] = (bf16)p;
                        l4[j2] = (bf16)(p - (float)h4[j2]);
                    }
                    *(bf16x4*)(Ph + q * LDP + key0 + 8 * jj) = h4;               // keys key0 + 8jj .. +3 of query q
                    *(bf16x4*)(Pl + q * LDP + key0 + 8 * jj) = l4;
                }
            }
        }
        SFB(10)
        lds_barrier();
        {
            PHASE_IDS;
            f32x16 O;
            SFB(11)
#pragma unroll
            for (int j = 0; j < 16; ++j) O[j] = 0.0f;
            const int q4 = (lane_p & 15) >> 2, p4 = lane_p & 3, g1_ = (lane_p >> 4) & 1;
            if (!skipw) {
                const int ksn = (mh == 0 || nrows <= 32) ? 2 : 4;        // keys 32.. : above the diagonal of queries < 32, or unused
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks >= ksn) continue;
                    const bf16x8 ph = *(const bf16x8*)(Ph + (32 * mh + r) * LDP + 16 * ks + 8 * hh);
                    const bf16x8 pl = *(const bf16x8*)(Pl + (32 * mh + r) * LDP + 16 * ks + 8 * hh);
                    const bf16* Vp = R0 + (16 * ks + 8 * hh + q4) * LDR + 16 * g1_ + 4 * p4 + 32 * nb;
                    const bf16x8 vh = cat4(tr_read(Vp), tr_read(Vp + 4 * LDR));
                    const bf16x8 vl = cat4(tr_read(Vp + TR * LDR), tr_read(Vp + TR * LDR + 4 * LDR));
                    O = mfma_bf16(pl, vh, O);
                    O = mfma_bf16(ph, vl, O);
                    O = mfma_bf16(ph, vh, O);
                }
            }
            load_bfrags((const bf16*)k.w[3], nb, r, hh, bh, bl);        // W1, consumed after LN2
            const Out o = make_rows(k.x1, mrow0, mrows, H);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
            float* Xp = Xf + t0 * XS + n;
            // ---- x1 = O + LN(x) (modules.py:223); the K tile is dead since the first barrier of the phase: R2 is Xf again
            if (!skipw) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float v = O[j] + qres[j];
                    Xp[ROWJ(j) * XS] = v;
                    bstore(o, row_base(pruned, t0 + ROWJ(j), H4, info_l) + n4, v);
                }
            }
        }
        }
        SFB(12)
        lds_barrier();
        // ---- LN2 (ADER.py:75): y -> memory, Xf (fp32, the FFN residual) and hi/lo -> R0
        {
            const Out oy = make_rows(k.y, mrow0, mrows, H);
            const Out om = make_rows(k.mean2, mrow0, mrows, 1), os = make_rows(k.std2, mrow0, mrows, 1);
            const int lane_q = opaque(lane);
            const int sub = lane_q & 15, rsub = lane_q >> 4;
#pragma unroll 1
            for (int pass = 0; pass < npass; ++pass) {
                const int t = 40 * pass + 4 * wave + rsub;
                if (t < nrows) {
                    float x[10], s = 0.0f;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int c = sub + 16 * i;
                        x[i] = (c < H) ? Xf[t * XS + c] : 0.0f;
                        s += x[i];
                    }
                    s = row16_sum(s);
                    const float mean = s / (float)H;
                    float q = 0.0f;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const float dlt = (sub + 16 * i < H) ? (x[i] - mean) : 0.0f;
                        q += dlt * dlt;
                    }
                    q = row16_sum(q);
                    const float sd = sqrtf(q / (float)H + LN_EPS);
                    const float rsd = 1.0f / sd;
                    bf16* T0 = R0 + t * LDR + sub;
                    const uint32_t bo = row_base(pruned, t, H4, info_l) + (uint32_t)sub * 4u;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int c = sub + 16 * i;
                        const float y = (c < H) ? g2[i] * ((x[i] - mean) * rsd) + be2[i] : 0.0f;
                        put_split(T0, T0 + TR * LDR, 16 * i, y);
                        Xf[t * XS + c] = y;
                        bstore(oy, (c < H) ? bo + 64u * i : OOB, y);
                    }
                    const uint32_t so = (sub == 0) ? row_base(pruned, t, 4u, info_l) : OOB;
                    bstore(om, so, mean); bstore(os, so, sd);
                }
            }
        }
        SFB(13)
        lds_barrier();
        if (SMALL) {
            const int lane_s = opaque(lane);
            const int c = lane_s & 15, g = lane_s >> 4;
            const int n = 16 * wave + c;
            const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
            // ---- h1 = dropout(relu(y.W1 + b1)) -> memory, hi/lo -> R1
            {
                f32x4v acc[2];
                tile_mma16(R0, c, g, bh, bl, nrb, acc);
                load_bfrags16((const bf16*)k.w[4], wave, lane_s, bh, bl);
                const uint32_t thr = k.d_ffn1.thr, key = k.d_ffn1.key;
                const float scale = thr ? k.d_ffn1.scale : 1.0f;
                const Out o = make_rows(k.h1d, mrow0, mrows, H);
                bf16* Th = R1 + (4 * g) * LDR + n;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    if (rb >= nrb) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = 16 * rb + 4 * g + i;
                        float v = fmaxf(acc[rb][i] + bias5[3], 0.0f);
                        if (thr) v = ((lowbias32((gph_l[t] + (uint32_t)n) ^ key) >> 8) >= thr) ? v * scale : 0.0f;
                        put_split(Th, Th + TR * LDR, (16 * rb + i) * LDR, v);
                        bstore(o, row_base(pruned, t, H4, info_l) + n4, v);
                    }
                }
            }
            lds_barrier();
            // ---- x2 = (dropout(h1.W2 + b2) + y) * (id != 0)
            {
                f32x4v acc[2];
                tile_mma16(R1, c, g, bh, bl, nrb, acc);
                if (l + 1 < a.L) load_bfrags16((const bf16*)kp[1].w[0], wave, lane_s, bh, bl);
                const uint32_t thr = k.d_ffn2.thr, key = k.d_ffn2.key;
                const float scale = thr ? k.d_ffn2.scale : 1.0f;
                const Out o = make_rows(k.x2, mrow0, mrows, H);
                float* Xp = Xf + (4 * g) * XS + n;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    if (rb >= nrb) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = 16 * rb + 4 * g + i;
                        float v = acc[rb][i] + bias5[4];
                        if (thr) v = ((lowbias32((gph_l[t] + (uint32_t)n) ^ key) >> 8) >= thr) ? v * scale : 0.0f;
                        const float yv = (n < XS) ? Xp[(16 * rb + i) * XS] : 0.0f;
                        v = (sq_l[t] != 0) ? v + yv : 0.0f;
                        if (n < XS) Xp[(16 * rb + i) * XS] = v;
                        bstore(o, row_base(pruned, t, H4, info_l) + n4, v);
                    }
                }
            }
        } else {
        // ---- h1 = dropout(relu(y.W1 + b1)) (modules.py:254-257) -> memory, hi/lo -> R1
        {
            PHASE_IDS;
            f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
            load_bfrags((const bf16*)k.w[4], nb, r, hh, bh, bl);
            const uint32_t thr = k.d_ffn1.thr, key = k.d_ffn1.key;
            const float scale = thr ? k.d_ffn1.scale : 1.0f;
            const Out o = make_rows(k.h1d, mrow0, mrows, H);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
            bf16* Th = R1 + t0 * LDR + n;
#define F1_EPI(DR_)                                                                                                \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                       \
                const int t = t0 + ROWJ(j);                                                                        \
                float v = fmaxf(acc[j] + bias5[3], 0.0f);                                                          \
                if (DR_) v = ((lowbias32((gph_l[t] + (uint32_t)n) ^ key) >> 8) >= thr) ? v * scale : 0.0f;         \
                put_split(Th, Th + TR * LDR, ROWJ(j) * LDR, v);                                                    \
                bstore(o, row_base(pruned, t, H4, info_l) + n4, v);                                                \
            }
            if (skipw) {
            } else if (thr) { F1_EPI(true) } else { F1_EPI(false) }
#undef F1_EPI
        }
        SFB(14)
        lds_barrier();
        // ---- x2 = (dropout(h1.W2 + b2) + y) * (id != 0) (modules.py:258-266, ADER.py:80)
        {
            PHASE_IDS;
            f32x16 acc = tile_mma(R1, mh, r, hh, bh, bl, skipw);
            if (l + 1 < a.L) load_bfrags((const bf16*)kp[1].w[0], nb, r, hh, bh, bl);
            const uint32_t thr = k.d_ffn2.thr, key = k.d_ffn2.key;
            const float scale = thr ? k.d_ffn2.scale : 1.0f;
            const Out o = make_rows(k.x2, mrow0, mrows, H);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
            float* Xp = Xf + t0 * XS + n;
#define F2_EPI(DR_)                                                                                                \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) {                                                       \
                const int t = t0 + ROWJ(j);                                                                        \
                float v = acc[j] + bias5[4];                                                                       \
                if (DR_) v = ((lowbias32((gph_l[t] + (uint32_t)n) ^ key) >> 8) >= thr) ? v * scale : 0.0f;         \
                const float yv = (n < XS) ? Xp[ROWJ(j) * XS] : 0.0f;                                               \
                v = (sq_l[t] != 0) ? v + yv : 0.0f;                                                                \
                if (n < XS) Xp[ROWJ(j) * XS] = v;                                                                  \
                bstore(o, row_base(pruned, t, H4, info_l) + n4, v);                                                \
            }
            if (skipw) {
            } else if (thr) { F2_EPI(true) } else { F2_EPI(false) }
#undef F2_EPI
        }
        }
        SFB(15)
        lds_barrier();
#undef k
    }
}

__global__ __launch_bounds__(640) void k_seqp_fwd(AderSeqFwd a, AderSeqPack pk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R0 = (bf16*)smem_raw;
    bf16* R1 = R0 + RSZ;
    bf16* R2 = R1 + RSZ;
    float* Xf = (float*)R2;
    float* km_l = (float*)(R2 + RSZ);            // [64] key mask of the current block
    float* qm_l = km_l + TR;                     // [64] query mask
    int* sq_l = (int*)(qm_l + TR);               // [64] item id of the row
    float* red_l = (float*)(sq_l + TR);          // [2][2][64] softmax max / sum halves
    int* info_l = (int*)(red_l + 4 * TR);        // [64] plan record: session start row | last << 6 | t << 8 | b << 16
    uint32_t* gph_l = (uint32_t*)(info_l + TR);  // [64] global position * H   (dropout counters of the row sites)
    uint32_t* gpt_l = gph_l + TR;                // [64] global position * T   (attention dropout counters)
    int* tp_l = (int*)(gpt_l + TR);              // [64] position t of the row
    const int tile = blockIdx.x;
    if (tile >= pk.hdr[0]) return;               // (the grid is the host's upper bound of the tile count)
    const int nrows = pk.tile_rows[tile];
    const size_t prow0 = (size_t)tile * TR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = wave % 5, mh = wave / 5;
    const int T = a.T, H = a.H;
    const uint32_t H4 = (uint32_t)H * 4u;
    SFS_INIT

    for (int i = tid; i < 2 * RSZ * (int)sizeof(bf16) / 16; i += 640) ((uint4*)R0)[i] = make_uint4(0u, 0u, 0u, 0u);   // R0, R1
    // ---- prologue (modules.py:118-130, ADER.py:41-60): x0 = dropout(E[id]*sqrt(H) + P[t]) * (id != 0)
    {
        const Out ox0 = make_rows(a.x0, prow0, nrows, H);
        const uint32_t thr = a.d_emb.thr, key = a.d_emb.key;
        const float scale = thr ? a.d_emb.scale : 1.0f;
        int ids[7];
        float ev[7][3], pv[7][3];
#pragma unroll
        for (int u = 0; u < 7; ++u) {                       // the wave's rows: all gathers in flight before any use
            const int r = wave + 10 * u;
            int id = 0, inf = r;                            // unused row: a one-row session of its own, id 0
            uint32_t gp = 0u;
            if (r < nrows) {
                id = pk.ids[prow0 + r]; inf = pk.info[prow0 + r]; gp = pk.gpos[prow0 + r];
                if (id < 0 || id >= a.V) {
                    if (lane == 0) atomicOr(a.status, ADER_ST_BAD_ID);
                    id = 0;
                }
            }
            ids[u] = id;
            const int tp = (inf >> 8) & 63;
            if (lane == 0 && r < TR) { sq_l[r] = id; info_l[r] = inf; gph_l[r] = gp * (uint32_t)H; gpt_l[r] = gp * (uint32_t)T; tp_l[r] = tp; }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = lane + 64 * i;
                pv[u][i] = (r < nrows && c < H) ? a.pos[(size_t)tp * H + c] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int r = wave + 10 * u;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = lane + 64 * i;
                ev[u][i] = (r < nrows && c < H && ids[u] != 0) ? a.emb[(size_t)ids[u] * H + c] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int r = wave + 10 * u;
            if (r < TR) {
                const uint32_t gh = gph_l[r];               // (written by this wave's lane 0 above: in order)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int c = lane + 64 * i;
                    float v = ev[u][i] * a.sqrtH + pv[u][i];
                    if (thr) v = ((lowbias32((gh + (uint32_t)c) ^ key) >> 8) >= thr) ? v * scale : 0.0f;
                    v = (ids[u] != 0 && r < nrows && c < H) ? v : 0.0f;
                    if (c < XS) Xf[r * XS + c] = v;
                    bstore(ox0, (c < H) ? (uint32_t)(r * H + c) * 4u : OOB, v);
                }
            }
        }
    }
    // tiles of at most 32 rows (the short class at the default window: nearly every tile of the shipped data) take the SMALL path: the
    // ten waves split the 160 output columns 16 each on v_mfma_f32_16x16x32_bf16 -- all ten work on the tile's rows, each with half the
    // MFMA clocks and half the epilogue of the 32x32 mapping, whose second row group (waves 5..9) would idle
    const bool small = nrows <= 32;
    const int nrb = nrows > 16 ? 2 : 1;               // 16-row blocks of the small path that hold rows
    lds_barrier();
    SFS(0)
    {
        SeqpCtx cx;
        cx.R0 = R0; cx.R1 = R1; cx.R2 = R2; cx.km_l = km_l; cx.qm_l = qm_l; cx.sq_l = sq_l; cx.red_l = red_l; cx.info_l = info_l;
        cx.gph_l = gph_l; cx.gpt_l = gpt_l; cx.tp_l = tp_l; cx.nrows = nrows; cx.prow0 = prow0; cx.nrb = nrb;
#ifdef SFP_STAMP
        cx.seg = seg; cx.tprev = &tprev;
#endif
        // (two instantiations of the block loop, not one loop with a branch per phase: with both mappings in one body hipcc spilled
        //  ~300 registers)
        if (small) seqp_blocks<true>(a, cx);
        else seqp_blocks<false>(a, cx);
    }
    SFS(33)
    // ---- final LayerNorm of every session's last position (ADER.py:83-85) -> rep[b]
    {
        float gf[3], bf_[3];
        const int lane_q = opaque(lane);
        load3(a.lnf_g, H, lane_q, gf); load3(a.lnf_b, H, lane_q, bf_);
#pragma unroll 1
        for (int r = wave; r < nrows; r += 10) {
            const int inf = info_l[r];
            if (!(inf & 64)) continue;
            const int b = inf >> 16;
            float x[3], y[3], mean, sd, xs, ys;
            ln_row(Xf + r * XS, true, H, lane, gf, bf_, x, y, mean, sd, xs, ys);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = lane + 64 * i;
                if (c < H) a.rep[(size_t)b * H + c] = y[i];
            }
            if (lane == 0) { a.meanf[b] = mean; a.stdf[b] = sd; }
        }
    }
    SFS(34)
    SFS_DUMP
}

static const size_t kSeqpFwdLds = (size_t)3 * RSZ * sizeof(bf16) + (size_t)11 * TR * sizeof(float);

extern "C" {

int ader_seqp_fwd(const AderSeqFwd* desc, const AderSeqPack* pack, int max_tiles, void* stream) {
    const AderSeqFwd& a = *desc;
    if (a.B <= 0) return 0;
    if (a.T < 1 || a.T > TR || a.H < 2 || a.H > 150 || (a.H & 1) || a.L < 1 || a.L > ADER_SEQ_MAXL || a.B > 4096 || !pack) return -2;
    if (max_tiles <= 0 || max_tiles > a.B) max_tiles = a.B;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_seqp_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSeqpFwdLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_seqp_fwd, dim3(max_tiles), dim3(640), kSeqpFwdLds, (hipStream_t)stream, a, *pack);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
