// Whole SASRec forward of one session in ONE workgroup (reference ADER.py:41-91 + modules.py: embedding prologue,
// num_blocks x [LayerNorm -> Q/K/V -> causal masked attention -> residual -> LayerNorm -> FFN -> residual -> mask],
// final LayerNorm of position T-1).  Every operation of the stack is local to a row or to a session, so a workgroup
// that owns the T <= 64 rows of a session never has to leave the CU: the residual stream and the GEMM / attention
// operands stay in LDS, the weights stream from L2, and only the activations the backward pass needs are written to
// memory (same buffers and layouts as the per-op kernels of rowwise/gemm_x3/attn_x3.hip, which this kernel replaces:
// ~24 dependent launches -> 1).  Arithmetic is the same "bf16x3" scheme: fp32 operands split into bf16 hi+lo, three
// v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation, fp32 LayerNorm / softmax / residuals.
//
// 10 waves; in GEMM and attention phases wave (mh = w/5, nb = w%5) owns rows 32mh.. and columns 32nb.. of the 64x160 tile,
// so a lane always holds element (row 32mh + acc_row(j,hh), col 32nb + r) in accumulator register j.
// LDS: three hi/lo tile pairs R0,R1,R2 of [64][168] bf16 (stride 336 B: conflict-free ds_read_b128); R2 doubles as the
// fp32 residual-stream tile Xf [64][164] while no K tile is live.
// heads == 1, T <= 64, H even and <= 150 (the engine falls back to the per-op kernels otherwise).  gfx950 only.
#include "seq_common.h"

#ifdef SF_STAMP     // diagnostic build only (tools/build_variant.sh ... -DSF_STAMP): clocks per phase of waves 0 and 9 of each workgroup
__device__ unsigned long long sf_dbg[2 * 40 * 1024];
#define SFS_INIT unsigned long long seg[40], tprev; for (int i_ = 0; i_ < 40; ++i_) seg[i_] = 0; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#define SFS(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                  __builtin_amdgcn_sched_barrier(0); seg[k_] += t_ - tprev; tprev = t_; }
#define SFB(k_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                  __builtin_amdgcn_sched_barrier(0); if (l == 0) seg[1 + k_] += t_ - tprev; else seg[17 + k_] += t_ - tprev; tprev = t_; }
#define SFS_DUMP { if (lane == 0 && (wave == 0 || wave == 9) && blockIdx.x < 1024) for (int k_ = 0; k_ < 40; ++k_) \
                       sf_dbg[(blockIdx.x * 2 + (wave == 9)) * 40 + k_] = seg[k_]; }
extern "C" int ader_dbg_read_sf(void* dst, int n) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sf_dbg), (size_t)n * 8); }
#else
#define SFS_INIT
#define SFS(k_)
#define SFB(k_)
#define SFS_DUMP {}
#endif

__global__ __launch_bounds__(640) void k_seq_fwd(AderSeqFwd a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R0 = (bf16*)smem_raw;
    bf16* R1 = R0 + RSZ;
    bf16* R2 = R1 + RSZ;
    float* Xf = (float*)R2;
    float* km_l = (float*)(R2 + RSZ);            // [64] key mask of the current block
    float* qm_l = km_l + TR;                     // [64] query mask
    int* sq_l = (int*)(qm_l + TR);               // [64] item ids of the session
    float* red_l = (float*)(sq_l + TR);          // [2][2][64] softmax max / sum halves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = wave % 5, mh = wave / 5;
    const int b = blockIdx.x, T = a.T, H = a.H;
    const uint32_t H4 = (uint32_t)H * 4u;
    const uint32_t didx_row0 = (uint32_t)b * (uint32_t)T * (uint32_t)H;      // dropout counter of element (row0, 0)
    bf16x8 bh[10], bl[10];
    SFS_INIT
    // per-block descriptors are indexed with a runtime l: read them straight from the kernarg segment (indexing the by-value
    // struct would make the compiler copy it to scratch)
    typedef const AderSeqBlock __attribute__((address_space(4))) * BlkPtr;
    const BlkPtr blks = (BlkPtr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
                                 offsetof(AderSeqFwd, blk));

    for (int i = tid; i < 2 * RSZ * (int)sizeof(bf16) / 16; i += 640) ((uint4*)R0)[i] = make_uint4(0u, 0u, 0u, 0u);   // R0, R1
    // ---- prologue (modules.py:118-130, ADER.py:41-60): x0 = dropout(E[seq]*sqrt(H) + P[t]) * (seq != 0)
    {
        const SDrop d0 = sdrop_of(a.d_emb, didx_row0);
        const Out ox0 = make_out(a.x0, b, T, H, false);
        int ids[7];
        float ev[7][3], pv[7][3];
#pragma unroll
        for (int u = 0; u < 7; ++u) {                       // the wave's rows: all gathers in flight before any use
            const int t = wave + 10 * u;
            int id = 0;
            if (t < T) {
                id = a.seq[(size_t)b * T + t];
                if (id < 0 || id >= a.V) {
                    if (lane == 0) atomicOr(a.status, ADER_ST_BAD_ID);
                    id = 0;
                }
            }
            ids[u] = id;
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int t = wave + 10 * u;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int c = lane + 64 * i;
                const bool ok = t < T && c < H;
                ev[u][i] = (ok && ids[u] != 0) ? a.emb[(size_t)ids[u] * H + c] : 0.0f;
                pv[u][i] = ok ? a.pos[(size_t)t * H + c] : 0.0f;
            }
        }
#define PRO_ROWS(DR_)                                                                                      \
        _Pragma("unroll") for (int u = 0; u < 7; ++u) {                                                    \
            const int t = wave + 10 * u;                                                                   \
            if (t < TR) {                                                                                  \
                if (lane == 0) sq_l[t] = ids[u];                                                           \
                _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                            \
                    const int c = lane + 64 * i;                                                           \
                    float v = ev[u][i] * a.sqrtH + pv[u][i];                                               \
                    v = sdrop_apply<DR_>(d0, didx_row0 + d0.off + (uint32_t)(t * H + c), v);               \
                    v = (ids[u] != 0 && t < T && c < H) ? v : 0.0f;                                        \
                    if (c < XS) Xf[t * XS + c] = v;                                                        \
                    bstore(ox0, (c < H) ? (uint32_t)(t * H + c) * 4u : OOB, v);                            \
                }                                                                                          \
            }                                                                                              \
        }
        if (d0.thr) { PRO_ROWS(true) } else { PRO_ROWS(false) }
#undef PRO_ROWS
    }
    {
        const int r = lane & 31, hh = lane >> 5;
        load_bfrags((const bf16*)blks[0].w[0], nb, r, hh, bh, bl);       // Wq of block 0 (in flight across the barrier)
    }
    lds_barrier();
    SFS(0)
    // ---- leading padding.  Sessions are left-padded (util.py:161-169); on real data ~90 % of the positions are padding.  A padded
    // position influences no real one -- its key is masked (modules.py:188-193), its outputs are re-zeroed (ADER.py:80) and its
    // gradient is exactly zero -- so rows [0, tv0) are skipped: no LayerNorm, no epilogue, no activation store (the backward
    // kernels skip the same rows and write their zero gradients).  Their tile rows keep stale but FINITE contents (the operand
    // tiles are cleared once above), so they only ever produce finite garbage in rows nobody consumes.
    int tv0;
    {
        const unsigned long long nz = __ballot(lane < T && sq_l[lane] != 0);
        tv0 = nz ? (int)__ffsll((long long)nz) - 1 : T;
    }
    const bool skipw = mh == 0 && tv0 >= 32;          // this wave's 32 rows are all leading padding

#pragma unroll 1
    for (int l = 0; l < a.L; ++l) {
        const BlkPtr kp = blks + l;
#define k (*kp)
        const bool pruned = k.pruned != 0;      // last block: only position T-1 of the query / FFN path is kept (ADER.py:85)
        // In a pruned block only row T-1 of the query / FFN path is consumed (rows are independent there; K and V still need every
        // row): the waves that do not own that row skip those products, and the owning waves run the epilogue for the ONE
        // accumulator register that holds it -- row T-1 = 32 mhT + 4 hhT + ROWJ(jT).  The other rows of the tiles keep stale
        // (finite or not: never consumed) contents.
        // small parameters of the block: requested now, consumed phases later
        float g1[10], be1[10], bias5[5];
        load10(k.ln1_g, H, lane & 15, g1); load10(k.ln1_b, H, lane & 15, be1);
        {
            const int n = 32 * nb + (lane & 31);
#pragma unroll
            for (int i = 0; i < 5; ++i) bias5[i] = (n < H) ? k.bias[i][n] : 0.0f;
        }
        // ---- LN1 (ADER.py:67 via modules.py:44-48) + key/query masks (modules.py:188,208); x -> R0, LN(x) -> R1
        //      16 lanes per row, 4 rows per wave at a time: lane (rsub, sub) owns columns sub + 16 i of row 4*(wave + 10 pass) + rsub
        {
            const Out oq = make_out(k.q_in, b, T, H, pruned);
            const Out om = make_out(k.mean1, b, T, 1, pruned), os = make_out(k.std1, b, T, 1, pruned);
            const Out okm = make_out(k.kmask, b, T, 1, false), oqm = make_out(k.qmask, b, T, 1, pruned);
            const int sub = lane & 15, rsub = lane >> 4;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                const int t = 40 * pass + 4 * wave + rsub;
                if (t < tv0) {                       // leading padding: only the masks exist (key mask 0)
                    if (sub == 0) { km_l[t] = 0.0f; qm_l[t] = 0.0f; }
                    bstore(okm, (sub == 0) ? (uint32_t)t * 4u : OOB, 0.0f);
                } else if (t < TR) {
                    const bool valid = t < T;
                    float x[10], s = 0.0f;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int c = sub + 16 * i;
                        x[i] = (valid && c < H) ? Xf[t * XS + c] : 0.0f;
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
                    const uint32_t bo = (uint32_t)(t * H + sub) * 4u;
#pragma unroll
                    for (int i = 0; i < 10; ++i) put_split(T0, T0 + TR * LDR, 16 * i, x[i]);
                    if (!pruned || t == T - 1) {        // LN(x) feeds the query path only
#pragma unroll
                        for (int i = 0; i < 10; ++i) {
                            const int c = sub + 16 * i;
                            const float y = (valid && c < H) ? g1[i] * ((x[i] - mean) * rsd) + be1[i] : 0.0f;
                            ys += y;
                            put_split(T1, T1 + TR * LDR, 16 * i, y);
                            bstore(oq, (c < H) ? bo + 64u * i : OOB, y);
                        }
                    }
                    ys = row16_sum(ys);
                    const float kmv = (valid && s != 0.0f) ? 1.0f : 0.0f, qmv = (valid && ys != 0.0f) ? 1.0f : 0.0f;
                    const uint32_t so = (sub == 0) ? (uint32_t)t * 4u : OOB;
                    if (sub == 0) { km_l[t] = kmv; qm_l[t] = qmv; }
                    bstore(okm, so, kmv); bstore(oqm, so, qmv); bstore(om, so, mean); bstore(os, so, sd);
                }
            }
        }
        SFB(0)
        lds_barrier();
        // ---- Q = LN(x).Wq + bq (modules.py:172) -> memory, hi/lo -> R1 (in place)
        {
            PHASE_IDS;
            PRUNE_IDS;
            SFB(1)
            f32x16 acc = tile_mma(R1, mh, r, hh, bh, bl, skipw);
            load_bfrags((const bf16*)k.w[1], nb, r, hh, bh, bl);
            SFB(2)
            const Out o = make_out(k.Q, b, T, H, pruned);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
            bf16* Th = R1 + t0 * LDR + n;
            lds_barrier();                                              // every wave has read its R1 rows
#define Q_EPI(accv_, rj_)                                                                                  \
            {                                                                                              \
                const float v = (n < H) ? (accv_) + bias5[0] : 0.0f;                                       \
                put_split(Th, Th + TR * LDR, (rj_) * LDR, v);                                              \
                bstore(o, boff0 + (rj_) * H4, v);                                                          \
            }
            if (skipw) {
            } else if (!pruned) {
#pragma unroll
                for (int j = 0; j < 16; ++j) Q_EPI(acc[j], ROWJ(j));
            } else if (act && hh == hhT) {
                Q_EPI(pick16(acc, jT), rjT);
            }
#undef Q_EPI
        }
        SFB(3)
        // ---- K = x.Wk + bk (modules.py:173) -> memory, hi/lo -> R2 (the fp32 tile is dead)
        {
            PHASE_IDS;
            f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
            load_bfrags((const bf16*)k.w[2], nb, r, hh, bh, bl);
            SFB(4)
            const Out o = make_out(k.K, b, T, H, false);
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
            const Out o = make_out(k.V, b, T, H, false);
            SFB(6)
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
            bf16* Th = R0 + t0 * LDR + n;
            lds_barrier();                                              // every wave has read its R0 rows
            if (!skipw) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float v = (n < H && t0 + ROWJ(j) < T) ? acc[j] + bias5[2] : 0.0f;   // rows >= T: exact zeros (0 * V below)
                    put_split(Th, Th + TR * LDR, ROWJ(j) * LDR, v);
                    bstore(o, boff0 + ROWJ(j) * H4, v);
                }
            }
        }
        SFB(7)
        __syncthreads();        // full barrier: LN(x) rows written to memory by other waves are re-read after the attention
        SFB(8)
        // ---- attention (modules.py:177-223).  Scores and softmax are computed ONCE per session by four waves -- wave (mq, kb)
        //      owns the 32x32 block S^T[keys 32kb..][queries 32mq..] (keys on the MFMA rows, the lane's query on the column, so a
        //      query's statistics are lane-local up to one exchange with the wave holding its other 32 keys) -- and the dropped
        //      probabilities go through a [query][key] hi/lo tile in LDS to all ten waves for O[:, 32nb..] = P_drop . V.
        float qres[16], g2[10], be2[10];
        load10(k.ln2_g, H, lane & 15, g2); load10(k.ln2_b, H, lane & 15, be2);      // consumed by LN2, after the attention
        {
            PHASE_IDS;
            const Out oq = make_out(k.q_in, b, T, H, pruned);
            const uint32_t boff0 = (n < H) ? (uint32_t)((32 * mh + 4 * hh) * H + n) * 4u : OOB;
#pragma unroll
            for (int j = 0; j < 16; ++j) qres[j] = bload(oq, boff0 + ROWJ(j) * H4);    // residual rows, added after P.V
        }
        SFB(9)
        bf16* Ph = R1;                                   // [64 queries][LDP] hi, then lo: overlays the Q tile once S is done
        bf16* Pl = R1 + TR * LDP;
        {
            PHASE_IDS;
            const int mq = wave >> 1, kb = wave & 1;
            const bool swave = wave < 4 && !(mq == 0 && tv0 >= 32);      // (queries 0..31 all leading padding: nobody reads their P rows)
            const int q = 32 * mq + r;
            const int key0 = 32 * kb + 4 * hh;           // key of register j: key0 + ROWJ(j)
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) S[j] = 0.0f;
            float mx = -INFINITY, sum = 0.0f;
            if (swave) {
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
                    if (km_l[key] == 0.0f) sc = NEG_PAD;                         // modules.py:188-193 (0 beyond T)
                    if (key > q) sc = NEG_PAD;                                   // modules.py:196-202
                    if (key < T) mx = fmaxf(mx, sc);
                    S[j] = sc;
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (hh == 0) red_l[kb * TR + q] = mx;
            }
            lds_barrier();                               // also: every read of the Q and K tiles is done
            if (swave) {
                mx = fmaxf(mx, red_l[(kb ^ 1) * TR + q]);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float e = (key0 + ROWJ(j) < T) ? expf(S[j] - mx) : 0.0f;
                    S[j] = e;
                    sum += e;
                }
                sum += __shfl_xor(sum, 32, 64);
                if (hh == 0) red_l[2 * TR + kb * TR + q] = sum;
            }
            lds_barrier();
            if (swave) {
                sum += red_l[2 * TR + (kb ^ 1) * TR + q];
                const float r_sum = 1.0f / sum;
                const float qm = (q < T) ? qm_l[q] : 0.0f;                        // modules.py:208-211
                const SDrop da = sdrop_of(k.d_attn, (uint32_t)b * (uint32_t)T * (uint32_t)T);
                // P^T [key][query] (pruned: the row of query T-1 only, [key])
                const Out op = make_out(k.P, b, T, pruned ? 1 : T, false);
                const uint32_t T4 = pruned ? 4u : (uint32_t)T * 4u;
                uint32_t poff0 = pruned ? ((q == T - 1) ? 0u : OOB) : ((q < T) ? (uint32_t)q * 4u : OOB);
                poff0 += (uint32_t)key0 * T4;
                const uint32_t dbase = (uint32_t)b * (uint32_t)T * (uint32_t)T + da.off + (uint32_t)(q * T + key0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    bf16x4 h4, l4;
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        const int j = 4 * jj + j2;
                        float p = S[j] * r_sum;
                        bstore(op, poff0 + (uint32_t)ROWJ(j) * T4, p);
                        p = sdrop_apply1(da, dbase + (uint32_t)ROWJ(j), p * qm);  // modules.py:214
                        p = (q < T && key0 + ROWJ(j) < T) ? p : 0.0f;
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
            PRUNE_IDS;
            if (act) {
            const int ks0 = (tv0 >= 32) ? 2 : 0;            // keys 0..31 all leading padding: their probabilities are exact zeros
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < ks0) continue;
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
            const Out o = make_out(k.x1, b, T, H, pruned);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
            float* Xp = Xf + t0 * XS + n;
            // ---- x1 = O + LN(x) (modules.py:223); the K tile is dead since the first barrier of the phase: R2 is Xf again
            if (skipw) {
            } else if (!pruned) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float v = O[j] + qres[j];
                    Xp[ROWJ(j) * XS] = v;
                    bstore(o, boff0 + ROWJ(j) * H4, v);
                }
            } else if (act && hh == hhT) {
                const float v = pick16(O, jT) + bload(make_out(k.q_in, b, T, H, true), boff0 + rjT * H4);
                Xp[rjT * XS] = v;
                bstore(o, boff0 + rjT * H4, v);
            }
        }
        SFB(12)
        lds_barrier();
        // ---- LN2 (ADER.py:75): y -> memory, Xf (fp32, the FFN residual) and hi/lo -> R0
        {
            const Out oy = make_out(k.y, b, T, H, pruned);
            const Out om = make_out(k.mean2, b, T, 1, pruned), os = make_out(k.std2, b, T, 1, pruned);
            const int sub = lane & 15, rsub = lane >> 4;
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                const int t = 40 * pass + 4 * wave + rsub;
                if (t < TR) {
                    const bool valid = t < T && (!pruned || t == T - 1);
                    if ((pruned && !valid) || t < tv0) continue;     // (16-lane row groups diverge; nothing downstream reads these rows)
                    float x[10], s = 0.0f;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int c = sub + 16 * i;
                        x[i] = (valid && c < H) ? Xf[t * XS + c] : 0.0f;
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
                    const uint32_t bo = (uint32_t)(t * H + sub) * 4u;
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int c = sub + 16 * i;
                        const float y = (valid && c < H) ? g2[i] * ((x[i] - mean) * rsd) + be2[i] : 0.0f;
                        put_split(T0, T0 + TR * LDR, 16 * i, y);
                        Xf[t * XS + c] = y;
                        bstore(oy, (c < H) ? bo + 64u * i : OOB, y);
                    }
                    const uint32_t so = (sub == 0) ? (uint32_t)t * 4u : OOB;
                    bstore(om, so, mean); bstore(os, so, sd);
                }
            }
        }
        SFB(13)
        lds_barrier();
        // ---- h1 = dropout(relu(y.W1 + b1)) (modules.py:254-257) -> memory, hi/lo -> R1
        {
            PHASE_IDS;
            PRUNE_IDS;
            f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
            load_bfrags((const bf16*)k.w[4], nb, r, hh, bh, bl);
            const SDrop d1 = sdrop_of(k.d_ffn1, didx_row0);
            const Out o = make_out(k.h1d, b, T, H, pruned);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
            const uint32_t didx0 = didx_row0 + d1.off + (uint32_t)(t0 * H + n);
            bf16* Th = R1 + t0 * LDR + n;
#define F1_EPI(accv_, rj_, DR_)                                                                               \
            {                                                                                              \
                /* columns >= H: zero weights and bias -> exactly 0; rows >= T: garbage nobody consumes */ \
                const float v = DR_(d1, didx0 + (rj_) * (uint32_t)H, fmaxf((accv_) + bias5[3], 0.0f));     \
                put_split(Th, Th + TR * LDR, (rj_) * LDR, v);                                              \
                bstore(o, boff0 + (rj_) * H4, v);                                                          \
            }
            if (skipw) {
            } else if (!pruned) {
                if (d1.thr) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) F1_EPI(acc[j], ROWJ(j), sdrop_apply<true>);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) F1_EPI(acc[j], ROWJ(j), sdrop_apply<false>);
                }
            } else if (act && hh == hhT) {
                F1_EPI(pick16(acc, jT), rjT, sdrop_apply1);
            }
#undef F1_EPI
        }
        SFB(14)
        lds_barrier();
        // ---- x2 = (dropout(h1.W2 + b2) + y) * (seq != 0) (modules.py:258-266, ADER.py:80)
        {
            PHASE_IDS;
            PRUNE_IDS;
            f32x16 acc = tile_mma(R1, mh, r, hh, bh, bl, skipw);
            if (l + 1 < a.L) load_bfrags((const bf16*)kp[1].w[0], nb, r, hh, bh, bl);
            const SDrop d2 = sdrop_of(k.d_ffn2, didx_row0);
            const Out o = make_out(k.x2, b, T, H, pruned);
            const int t0 = 32 * mh + 4 * hh;
            const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
            const uint32_t didx0 = didx_row0 + d2.off + (uint32_t)(t0 * H + n);
            float* Xp = Xf + t0 * XS + n;
#define F2_EPI(accv_, rj_, DR_)                                                                               \
            {                                                                                              \
                const int t = t0 + (rj_);                                                                  \
                float v = DR_(d2, didx0 + (rj_) * (uint32_t)H, (accv_) + bias5[4]);                        \
                const float yv = (n < XS) ? Xp[(rj_) * XS] : 0.0f;                                         \
                v = (sq_l[t] != 0) ? v + yv : 0.0f;                                                        \
                if (n < XS) Xp[(rj_) * XS] = v;                                                            \
                bstore(o, boff0 + (rj_) * H4, v);                                                          \
            }
            if (skipw) {
            } else if (!pruned) {
                if (d2.thr) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) F2_EPI(acc[j], ROWJ(j), sdrop_apply<true>);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) F2_EPI(acc[j], ROWJ(j), sdrop_apply<false>);
                }
            } else if (act && hh == hhT) {
                F2_EPI(pick16(acc, jT), rjT, sdrop_apply1);
            }
#undef F2_EPI
        }
        SFB(15)
        lds_barrier();
#undef k
    }
    SFS(33)
    // ---- final LayerNorm of position T-1 (ADER.py:83-85) -> rep[b]
    if (wave == 0) {
        float x[3], y[3], mean, sd, xs, ys, gf[3], bf_[3];
        load3(a.lnf_g, H, lane, gf); load3(a.lnf_b, H, lane, bf_);
        ln_row(Xf + (T - 1) * XS, true, H, lane, gf, bf_, x, y, mean, sd, xs, ys);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = lane + 64 * i;
            if (c < H) a.rep[(size_t)b * H + c] = y[i];
        }
        if (lane == 0) { a.meanf[b] = mean; a.stdf[b] = sd; }
    }
    SFS(34)
    SFS_DUMP
}

static const size_t kSeqFwdLds = (size_t)3 * RSZ * sizeof(bf16) + (size_t)7 * TR * sizeof(float);

extern "C" {

int ader_seq_fwd(const AderSeqFwd* desc, void* stream) {
    const AderSeqFwd& a = *desc;
    if (a.B <= 0) return 0;
    if (a.T < 1 || a.T > TR || a.H < 2 || a.H > 150 || (a.H & 1) || a.L < 1 || a.L > ADER_SEQ_MAXL) return -2;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_seq_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSeqFwdLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_seq_fwd, dim3(a.B), dim3(640), kSeqFwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
