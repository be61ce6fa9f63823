// Backward of one SASRec block on PACKED session tiles (tf.gradients of the graph built at ADER.py:62-81): the packed forms of
// k_seq_bwd_ffn / k_seq_bwd_qkv (seq_bwd.hip), k_attn_x3_bwd (attn_x3.hip) and k_attn_last_bwd (attn.hip).  One workgroup per
// tile of ader_seq_pack_plan (seqp_plan.hip); tensors in tile order as k_seqp_fwd (seqp_fwd.hip) wrote them; a pruned (last) block
// keeps its query / FFN path in the compact [B,..] tensors of the unpacked path (row b = the last position of session b).
// Same arithmetic as the unpacked kernels (bf16 hi/lo operands, three MFMAs per product, fp32 row math); what differs is the
// summation order of the per-tile partial sums (LayerNorm gamma / beta, per tile instead of per session) and of the attention
// products (a session's keys sit at other tile rows).  Rows of a tile beyond its row count are never written: the weight-gradient
// products skip them (ader_gemm_atb_x3_batch_pk takes the tile row counts).
// The gradient rows of the input embeddings (block 0, emb_bwd) leave the packed world: they are written to the session-indexed
// [B*T,H] tensor the table update, the positional gradient and the data-parallel exchanges address by position -- real positions
// only (every consumer goes through the id lists, which leave padding out, or through ader_pos_grad_packed).
// heads == 1, T <= 64, H even <= 150.  gfx950 only.
#include "seqp_common.h"

#define RED_FLOATS (10 * 2 * HP)          // per-wave gamma/beta partials

// dgamma/dbeta partials of the waves -> slab[tile][2][H] (fixed order: lanes, then waves 0..9)
__device__ __forceinline__ void flush_ln_partials_pk(float (&dg)[10], float (&db)[10], float* red_l, float* __restrict__ slab, int tile, int H,
                                                     int tid) {
    const int lane = tid & 63, wave = tid >> 6, sub = lane & 15;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        dg[i] += __shfl_xor(dg[i], 16, 64); dg[i] += __shfl_xor(dg[i], 32, 64);
        db[i] += __shfl_xor(db[i], 16, 64); db[i] += __shfl_xor(db[i], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            red_l[(wave * 2 + 0) * HP + sub + 16 * i] = dg[i];
            red_l[(wave * 2 + 1) * HP + sub + 16 * i] = db[i];
        }
    }
    lds_barrier();
    if (tid < 2 * HP) {
        const int which = tid / HP, c = tid - which * HP;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 10; ++w) s += red_l[(w * 2 + which) * HP + c];
        if (c < H) slab[((size_t)tile * 2 + which) * H + c] = s;
    }
}

// LayerNorm backward of the tile's rows (reference LN: modules.py:44-48).  dy rows in Xf (fp32 LDS), x rows and statistics in memory
// (xpr / spr: compact tensors of a pruned block).  Rows that do not exist in a pruned block (not a last position) get dx = 0.
template <bool STORE>
__device__ __forceinline__ void ln_bwd_rows_pk(float* Xf, const Out& ox, bool xpr, const Out& omean, const Out& ostd, bool spr,
                                               const float (&gam)[10], const Out& odx, int nrows, int npass, int H, bool pruned,
                                               const int* info_l, int wave, int lane, float (&dg)[10], float (&db)[10]) {
    const int sub = lane & 15, rsub = lane >> 4;
    const uint32_t H4 = (uint32_t)H * 4u;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        const int t = 40 * pass + 4 * wave + rsub;
        if (t >= nrows) continue;
        const bool valid = !pruned || (info_l[t] & 64);
        if (!valid) {
#pragma unroll
            for (int i = 0; i < 10; ++i) Xf[t * XS + sub + 16 * i] = 0.0f;
            continue;
        }
        const uint32_t bx = row_base(xpr, t, H4, info_l) + (uint32_t)sub * 4u;
        const uint32_t bs = row_base(spr, t, 4u, info_l);
        float xv[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) xv[i] = bload(ox, (sub + 16 * i < H) ? bx + 64u * i : OOB);
        const float mean = bload(omean, bs);
        const float sd = bload(ostd, bs);
        const float rsd = 1.0f / sd;                        // one reciprocal per row instead of 20 divisions
        float xh[10], dxh[10], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = sub + 16 * i;
            const float g = (c < H) ? Xf[t * XS + c] : 0.0f;
            xh[i] = (c < H) ? (xv[i] - mean) * rsd : 0.0f;
            dxh[i] = g * gam[i];
            s1 += dxh[i];
            s2 += dxh[i] * xh[i];
            dg[i] += g * xh[i];
            db[i] += g;
        }
        s1 = row16_sum(s1) / (float)H;
        s2 = row16_sum(s2) / (float)H;
        const uint32_t bd = STORE ? row_base(pruned, t, H4, info_l) + (uint32_t)sub * 4u : 0u;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = sub + 16 * i;
            const float o = (c < H) ? (dxh[i] - s1 - xh[i] * s2) * rsd : 0.0f;
            Xf[t * XS + c] = o;
            if (STORE) bstore(odx, (c < H) ? bd + 64u * i : OOB, o);
        }
    }
}

// rows of a [.,H] fp32 tensor (tile rows, or the compact rows of a pruned block: absent rows read as 0) -> hi/lo tile
__device__ __forceinline__ void stage_rows_pk(const Out& src, bool pr, bf16* R, int nrows, int npass, int H, const int* info_l, int wave,
                                              int lane) {
    const int sub = lane & 15, rsub = lane >> 4;
    const uint32_t H4 = (uint32_t)H * 4u;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        const int t = 40 * pass + 4 * wave + rsub;
        if (t < nrows) {
            const uint32_t bo = row_base(pr, t, H4, info_l) + (uint32_t)sub * 4u;
            float v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) v[i] = bload(src, (sub + 16 * i < H) ? bo + 64u * i : OOB);
            bf16* Tp = R + t * LDR + sub;
#pragma unroll
            for (int i = 0; i < 10; ++i) put_split(Tp, Tp + TR * LDR, 16 * i, v[i]);
        }
    }
}

template <bool SMALL>
__device__ __forceinline__ void seqp_bwd_ffn_body(const AderSeqBwdFfn& a, const AderSeqPack& pk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R0 = (bf16*)smem_raw;
    bf16* R1 = R0 + RSZ;
    float* Xf = (float*)(R1 + RSZ);
    float* red_l = (float*)((bf16*)Xf + RSZ);
    int* sq_l = (int*)(red_l + RED_FLOATS);
    int* info_l = sq_l + TR;
    uint32_t* gph_l = (uint32_t*)(info_l + TR);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, H = a.H;
    const int nrows = pk.tile_rows[tile];
    const size_t prow0 = (size_t)tile * TR;
    const int nb = wave % 5, mh = wave / 5;
    const bool pruned = a.pruned != 0;
    const int mrows = pruned ? a.B : nrows;
    const size_t mrow0 = pruned ? 0 : prow0;
    const uint32_t H4 = (uint32_t)H * 4u;
    bf16x8 bh[10], bl[10];
    const int nrb = nrows > 16 ? 2 : 1;
    if (SMALL) load_bfrags16((const bf16*)a.w2 + 2 * WSZ, wave, lane, bh, bl);      // W2 planes [n][k] = W2[n][k]: A . W2^T
    else load_bfrags((const bf16*)a.w2 + 2 * WSZ, nb, lane & 31, lane >> 5, bh, bl);
    if (tid < TR) {
        const bool ok = tid < nrows;
        sq_l[tid] = ok ? pk.ids[prow0 + tid] : 0;
        info_l[tid] = ok ? pk.info[prow0 + tid] : tid;
        gph_l[tid] = ok ? pk.gpos[prow0 + tid] * (uint32_t)H : 0u;
    }
    for (int i = tid; i < RSZ * (int)sizeof(bf16) / 16; i += 640) ((uint4*)R0)[i] = make_uint4(0u, 0u, 0u, 0u);   // rows beyond nrows: zeros
    float gam[10];
    load10(a.ln2_g, H, lane & 15, gam);
    lds_barrier();
    const bool skipw = mh == 1 && nrows <= 32;
    const int npass = nrows > 40 ? 2 : 1;
    // ---- g = dx2 * (id != 0) -> Xf;  dh2 = g * keep * scale (modules.py:262-266, ADER.py:80) -> memory, hi/lo -> R0
    {
        const Out odx = make_rows(a.dx2, mrow0, mrows, H), odh = make_rows(a.dh2, mrow0, mrows, H);
        const uint32_t thr = a.d_ffn2.thr, key = a.d_ffn2.key;
        const float scale = thr ? a.d_ffn2.scale : 1.0f;
        const int sub = lane & 15, rsub = lane >> 4;
#pragma unroll 1
        for (int pass = 0; pass < npass; ++pass) {
            const int t = 40 * pass + 4 * wave + rsub;
            if (t < nrows) {
                const uint32_t bo = row_base(pruned, t, H4, info_l) + (uint32_t)sub * 4u;
                float v[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) v[i] = bload(odx, (sub + 16 * i < H) ? bo + 64u * i : OOB);
                const bool live = sq_l[t] != 0;
                const uint32_t gh = gph_l[t];
                bf16* Tp = R0 + t * LDR + sub;
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int c = sub + 16 * i;
                    const float g = live ? v[i] : 0.0f;
                    Xf[t * XS + c] = g;
                    float dh = g;
                    if (thr) dh = ((lowbias32((gh + (uint32_t)c) ^ key) >> 8) >= thr) ? g * scale : 0.0f;
                    put_split(Tp, Tp + TR * LDR, 16 * i, dh);
                    bstore(odh, (c < H) ? bo + 64u * i : OOB, dh);
                }
            }
        }
    }
    lds_barrier();
    if (SMALL) {
        // (16-column mapping of tiles with at most 32 rows: seqp_common.h)
        const int lane_s = opaque(lane);
        const int c = lane_s & 15, g = lane_s >> 4;
        const int n = 16 * wave + c;
        const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
        // ---- da = (dh2 . W2^T) * relu/dropout-grad -> memory, hi/lo -> R1
        {
            const Out oh = make_rows(a.h1d, mrow0, mrows, H), oda = make_rows(a.da, mrow0, mrows, H);
            float h1[8];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i) h1[4 * rb + i] = bload(oh, row_base(pruned, 16 * rb + 4 * g + i, H4, info_l) + n4);
            f32x4v acc[2];
            tile_mma16(R0, c, g, bh, bl, nrb, acc);
            load_bfrags16((const bf16*)a.w1 + 2 * WSZ, wave, lane_s, bh, bl);
            bf16* Th = R1 + (4 * g) * LDR + n;
            const float sc1 = a.d_ffn1.scale;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                if (rb >= nrb) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = (h1[4 * rb + i] != 0.0f) ? acc[rb][i] * sc1 : 0.0f;
                    put_split(Th, Th + TR * LDR, (16 * rb + i) * LDR, v);
                    bstore(oda, row_base(pruned, 16 * rb + 4 * g + i, H4, info_l) + n4, v);
                }
            }
        }
        lds_barrier();
        // ---- dy = da . W1^T + g  (in place in Xf)
        {
            f32x4v acc[2];
            tile_mma16(R1, c, g, bh, bl, nrb, acc);
            float* Xp = Xf + (4 * g) * XS + n;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                if (rb >= nrb) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) Xp[(16 * rb + i) * XS] += acc[rb][i];
            }
        }
        lds_barrier();
    } else {
    // ---- da = (dh2 . W2^T) * relu/dropout-grad (modules.py:254-257) -> memory, hi/lo -> R1
    {
        PHASE_IDS;
        const Out oh = make_rows(a.h1d, mrow0, mrows, H), oda = make_rows(a.da, mrow0, mrows, H);
        const int t0 = 32 * mh + 4 * hh;
        const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
        float h1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) h1[j] = bload(oh, row_base(pruned, t0 + ROWJ(j), H4, info_l) + n4);
        f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
        load_bfrags((const bf16*)a.w1 + 2 * WSZ, nb, r, hh, bh, bl);
        bf16* Th = R1 + t0 * LDR + n;
        const float sc1 = a.d_ffn1.scale;
        if (!skipw) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = (h1[j] != 0.0f) ? acc[j] * sc1 : 0.0f;
                put_split(Th, Th + TR * LDR, ROWJ(j) * LDR, v);
                bstore(oda, row_base(pruned, t0 + ROWJ(j), H4, info_l) + n4, v);
            }
        }
    }
    lds_barrier();
    // ---- dy = da . W1^T + g  (in place in Xf)
    {
        PHASE_IDS;
        f32x16 acc = tile_mma(R1, mh, r, hh, bh, bl, skipw);
        float* Xp = Xf + (32 * mh + 4 * hh) * XS + n;
        if (!skipw) {
#pragma unroll
            for (int j = 0; j < 16; ++j) Xp[ROWJ(j) * XS] += acc[j];
        }
    }
    lds_barrier();
    }
    // ---- LayerNorm2 backward -> dx1 (memory) + gamma/beta partials of the tile
    float dg[10], db[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { dg[i] = 0.0f; db[i] = 0.0f; }
    {
        const Out ox = make_rows(a.x1, mrow0, mrows, H), om = make_rows(a.mean2, mrow0, mrows, 1), os = make_rows(a.std2, mrow0, mrows, 1);
        const Out odx1 = make_rows(a.dx1, mrow0, mrows, H);
        ln_bwd_rows_pk<true>(Xf, ox, pruned, om, os, pruned, gam, odx1, nrows, npass, H, pruned, info_l, wave, lane, dg, db);
    }
    flush_ln_partials_pk(dg, db, red_l, a.slab, tile, H, tid);
}

__global__ __launch_bounds__(640) void k_seqp_bwd_ffn(AderSeqBwdFfn a, AderSeqPack pk) {
    const int tile = blockIdx.x, tid = threadIdx.x;
    if (tile >= pk.hdr[0]) {                 // (the grid is the host's bound of the tile count: the slab reduction sums every slot)
        if (tid < 2 * a.H) a.slab[(size_t)tile * 2 * a.H + tid] = 0.0f;
        return;
    }
    // (two instantiations, not a branch per phase: see k_seqp_fwd)
    if (pk.tile_rows[tile] <= 32) seqp_bwd_ffn_body<true>(a, pk);
    else seqp_bwd_ffn_body<false>(a, pk);
}

template <bool SMALL>
__device__ __forceinline__ void seqp_bwd_qkv_body(const AderSeqBwdQkv& a, const AderSeqPack& pk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R0 = (bf16*)smem_raw;
    bf16* R1 = R0 + RSZ;
    float* Xf = (float*)(R1 + RSZ);
    float* red_l = (float*)((bf16*)Xf + RSZ);
    int* sq_l = (int*)(red_l + RED_FLOATS);
    int* info_l = sq_l + TR;
    uint32_t* gph_l = (uint32_t*)(info_l + TR);
    int* lp_l = (int*)(gph_l + TR);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, H = a.H;
    const int nrows = pk.tile_rows[tile];
    const size_t prow0 = (size_t)tile * TR;
    const int nb = wave % 5, mh = wave / 5;
    const bool pruned = a.pruned != 0;
    const int mrows = pruned ? a.B : nrows;
    const size_t mrow0 = pruned ? 0 : prow0;
    const uint32_t H4 = (uint32_t)H * 4u;
    bf16x8 bh[10], bl[10];
    const int nrb = nrows > 16 ? 2 : 1;
    if (SMALL) load_bfrags16((const bf16*)a.wq + 2 * WSZ, wave, lane, bh, bl);
    else load_bfrags((const bf16*)a.wq + 2 * WSZ, nb, lane & 31, lane >> 5, bh, bl);
    if (tid < TR) {
        const bool ok = tid < nrows;
        sq_l[tid] = ok ? pk.ids[prow0 + tid] : 0;
        info_l[tid] = ok ? pk.info[prow0 + tid] : tid;
        gph_l[tid] = ok ? pk.gpos[prow0 + tid] * (uint32_t)H : 0u;
        lp_l[tid] = ok ? pk.lpos[prow0 + tid] : -1;
    }
    for (int i = tid; i < 2 * RSZ * (int)sizeof(bf16) / 16; i += 640) ((uint4*)R0)[i] = make_uint4(0u, 0u, 0u, 0u);   // R0, R1
    const bool skipw = mh == 1 && nrows <= 32;
    const int npass = nrows > 40 ? 2 : 1;
    float gam[10];
    load10(a.ln1_g, H, lane & 15, gam);
    lds_barrier();
    // ---- dQ rows -> R0 (a pruned block has the rows of the last positions only), dK rows -> R1
    {
        const Out oq = make_rows(a.dQ, mrow0, mrows, H);
        stage_rows_pk(oq, pruned, R0, nrows, npass, H, info_l, wave, lane);
        const Out ok = make_rows(a.dK, prow0, nrows, H);
        stage_rows_pk(ok, false, R1, nrows, npass, H, info_l, wave, lane);
    }
    lds_barrier();
    if (SMALL) {
        const int lane_s = opaque(lane);
        const int c = lane_s & 15, g = lane_s >> 4;
        const int n = 16 * wave + c;
        const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
        // ---- dqin = dQ . Wq^T + dx1 -> Xf
        const Out ox1 = make_rows(a.dx1, mrow0, mrows, H);
        float res[8];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) res[4 * rb + i] = bload(ox1, row_base(pruned, 16 * rb + 4 * g + i, H4, info_l) + n4);
        f32x4v acc[2];
        tile_mma16(R0, c, g, bh, bl, nrb, acc);
        load_bfrags16((const bf16*)a.wk + 2 * WSZ, wave, lane_s, bh, bl);
        float* Xp = Xf + (4 * g) * XS + n;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (rb >= nrb) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) Xp[(16 * rb + i) * XS] = acc[rb][i] + res[4 * rb + i];
        }
        lds_barrier();
    } else {
    // ---- dqin = dQ . Wq^T + dx1 -> Xf
    {
        PHASE_IDS;
        const Out ox1 = make_rows(a.dx1, mrow0, mrows, H);
        const int t0 = 32 * mh + 4 * hh;
        const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
        float res[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) res[j] = bload(ox1, row_base(pruned, t0 + ROWJ(j), H4, info_l) + n4);
        f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
        load_bfrags((const bf16*)a.wk + 2 * WSZ, nb, r, hh, bh, bl);
        float* Xp = Xf + t0 * XS + n;
        if (!skipw) {
#pragma unroll
            for (int j = 0; j < 16; ++j) Xp[ROWJ(j) * XS] = acc[j] + res[j];
        }
    }
    lds_barrier();
    }
    // ---- dV rows -> R0 (dQ is consumed); LayerNorm1 backward in place in Xf + gamma/beta partials
    float dg[10], db[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { dg[i] = 0.0f; db[i] = 0.0f; }
    {
        const Out ov = make_rows(a.dV, prow0, nrows, H);
        stage_rows_pk(ov, false, R0, nrows, npass, H, info_l, wave, lane);
        // x rows of the block input are in tile order also for a pruned block; its statistics are compact
        const Out ox = make_rows(a.x, prow0, nrows, H);
        const Out om = make_rows(a.mean1, mrow0, mrows, 1), os = make_rows(a.std1, mrow0, mrows, 1);
        ln_bwd_rows_pk<false>(Xf, ox, false, om, os, pruned, gam, ox, nrows, npass, H, pruned, info_l, wave, lane, dg, db);
    }
    lds_barrier();
    if (SMALL) {
        // ---- dx = LN1-backward + dK . Wk^T + dV . Wv^T  [block 0: * (id != 0) * keep * scale of the embedding prologue, rows stored by position]
        const int lane_s = opaque(lane);
        const int c = lane_s & 15, g = lane_s >> 4;
        const int n = 16 * wave + c;
        const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
        f32x4v acc[2];
        tile_mma16(R1, c, g, bh, bl, nrb, acc);
        load_bfrags16((const bf16*)a.wv + 2 * WSZ, wave, lane_s, bh, bl);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (rb >= nrb) continue;
            const bf16* Ah = R0 + (16 * rb + c) * LDR + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const bf16x8 ah = *(const bf16x8*)(Ah + 32 * ks);
                const bf16x8 al = *(const bf16x8*)(Ah + TR * LDR + 32 * ks);
                acc[rb] = mfma16_bf16(al, bh[ks], acc[rb]);
                acc[rb] = mfma16_bf16(ah, bl[ks], acc[rb]);
                acc[rb] = mfma16_bf16(ah, bh[ks], acc[rb]);
            }
        }
        const bool eb = a.emb_bwd != 0;
        const Out odx = eb ? make_rows(a.dx, 0, a.B * a.T, H) : make_rows(a.dx, prow0, nrows, H);
        const uint32_t thr = a.d_emb.thr, key = a.d_emb.key;
        const float scale = thr ? a.d_emb.scale : 1.0f;
        const float* Xp = Xf + (4 * g) * XS + n;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (rb >= nrb) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * rb + 4 * g + i;
                float v = Xp[(16 * rb + i) * XS] + acc[rb][i];
                uint32_t rb_ = (uint32_t)t * H4;
                if (eb) {
                    if (thr) v = ((lowbias32((gph_l[t] + (uint32_t)n) ^ key) >> 8) >= thr) ? v * scale : 0.0f;
                    v = (sq_l[t] != 0) ? v : 0.0f;
                    const int lp = lp_l[t];
                    rb_ = (lp >= 0) ? (uint32_t)lp * H4 : OOBH;
                }
                bstore(odx, rb_ + n4, v);
            }
        }
    } else {
    // ---- dx = LN1-backward + dK . Wk^T + dV . Wv^T  [block 0: * (id != 0) * keep * scale of the embedding prologue, rows stored by position]
    {
        PHASE_IDS;
        f32x16 acc = tile_mma(R1, mh, r, hh, bh, bl, skipw);
        load_bfrags((const bf16*)a.wv + 2 * WSZ, nb, r, hh, bh, bl);
        if (!skipw) {
            const bf16* Ah = R0 + (32 * mh + r) * LDR + 8 * hh;
            const bf16* Al = Ah + TR * LDR;
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                const bf16x8 ah = *(const bf16x8*)(Ah + 16 * ks);
                const bf16x8 al = *(const bf16x8*)(Al + 16 * ks);
                acc = mfma_bf16(al, bh[ks], acc);
                acc = mfma_bf16(ah, bl[ks], acc);
                acc = mfma_bf16(ah, bh[ks], acc);
            }
        }
        const bool eb = a.emb_bwd != 0;
        const Out odx = eb ? make_rows(a.dx, 0, a.B * a.T, H) : make_rows(a.dx, prow0, nrows, H);
        const uint32_t thr = a.d_emb.thr, key = a.d_emb.key;
        const float scale = thr ? a.d_emb.scale : 1.0f;
        const int t0 = 32 * mh + 4 * hh;
        const uint32_t n4 = (n < H) ? (uint32_t)n * 4u : OOBH;
        const float* Xp = Xf + t0 * XS + n;
        if (!skipw) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int t = t0 + ROWJ(j);
                float v = Xp[ROWJ(j) * XS] + acc[j];
                uint32_t rb = (uint32_t)t * H4;
                if (eb) {
                    if (thr) v = ((lowbias32((gph_l[t] + (uint32_t)n) ^ key) >> 8) >= thr) ? v * scale : 0.0f;
                    v = (sq_l[t] != 0) ? v : 0.0f;
                    const int lp = lp_l[t];
                    rb = (lp >= 0) ? (uint32_t)lp * H4 : OOBH;
                }
                bstore(odx, rb + n4, v);
            }
        }
    }
    }
    flush_ln_partials_pk(dg, db, red_l, a.slab, tile, H, tid);
}

__global__ __launch_bounds__(640) void k_seqp_bwd_qkv(AderSeqBwdQkv a, AderSeqPack pk) {
    const int tile = blockIdx.x, tid = threadIdx.x;
    if (tile >= pk.hdr[0]) {
        if (tid < 2 * a.H) a.slab[(size_t)tile * 2 * a.H + tid] = 0.0f;
        return;
    }
    if (pk.tile_rows[tile] <= 32) seqp_bwd_qkv_body<true>(a, pk);
    else seqp_bwd_qkv_body<false>(a, pk);
}

// ------------------------------------------------------------------------------------------------ attention backward, packed
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct AttnPkArgs {
    const float *Q, *K, *V, *dO;           // tile order [., H]
    const float *kmask, *qmask;            // tile order
    const float* PT;                        // [tile][key][query]
    float *dQ, *dK, *dV;
    int H, T;
    float sqrt_dh;
    uint32_t key, thr;
    float scale;
};

template <int NT>
__device__ __forceinline__ void stage_split_pk(bf16* Th, bf16* Tl, const float* __restrict__ src, int rows, int H, int tid) {
    const int HH = H >> 1, n2 = rows * HH;
    for (int i0 = 0; i0 < n2; i0 += NT * 8) {
        float2 v[8];
        int t[8], c2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i0 + tid + NT * u;
            t[u] = idx / HH; c2[u] = idx - t[u] * HH;
            v[u] = (idx < n2) ? *(const float2*)(src + (size_t)t[u] * H + 2 * c2[u]) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + tid + NT * u < n2) {
                bf16x2 h, l;
                h[0] = (bf16)v[u].x; h[1] = (bf16)v[u].y;
                l[0] = (bf16)(v[u].x - (float)h[0]); l[1] = (bf16)(v[u].y - (float)h[1]);
                *(bf16x2*)(Th + t[u] * LDR + 2 * c2[u]) = h;
                *(bf16x2*)(Tl + t[u] * LDR + 2 * c2[u]) = l;
            }
        }
    }
}
__device__ __forceinline__ void row_frags_pk(const float* __restrict__ src, bool valid, int dh, int hh, bf16x8 (&fh)[10], bf16x8 (&fl)[10]) {
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            const int k = 16 * ks + 8 * hh + 2 * j2;
            float2 v = make_float2(0.f, 0.f);
            if (valid && k < dh) v = *(const float2*)(src + k);
            const bf16 h0 = (bf16)v.x, h1 = (bf16)v.y;
            fh[ks][2 * j2] = h0; fh[ks][2 * j2 + 1] = h1;
            fl[ks][2 * j2] = (bf16)(v.x - (float)h0); fl[ks][2 * j2 + 1] = (bf16)(v.y - (float)h1);
        }
    }
}
__device__ __forceinline__ void rows_times_frags_pk(const bf16* Th, const bf16* Tl, const bf16x8 (&fh)[10], const bf16x8 (&fl)[10],
                                                    int ksteps, int nkb, int r, int hh, f32x16 (&acc)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[kb][j] = 0.0f;
        if (kb >= nkb) continue;
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            if (ks < ksteps) {
                const bf16x8 ah = *(const bf16x8*)(Th + (32 * kb + r) * LDR + 16 * ks + 8 * hh);
                const bf16x8 al = *(const bf16x8*)(Tl + (32 * kb + r) * LDR + 16 * ks + 8 * hh);
                acc[kb] = mfma_bf16(al, fh[ks], acc[kb]);
                acc[kb] = mfma_bf16(ah, fl[ks], acc[kb]);
                acc[kb] = mfma_bf16(ah, fh[ks], acc[kb]);
            }
        }
    }
}
__device__ __forceinline__ void acc_times_rows_pk(const f32x16 (&x)[2], const bf16* Th, const bf16* Tl, int nblocks, int nkb, int lane,
                                                  f32x16 (&O)[5], int nb0) {
    const int hh = lane >> 5, q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (kb >= nkb) continue;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = x[kb][8 * s + j];
                ph[j] = (bf16)v;
                pl[j] = (bf16)(v - (float)ph[j]);
            }
            const int ro = (32 * kb + 16 * s + 4 * hh + q4) * LDR + 16 * g1 + 4 * p4;
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                if (nb >= nb0 && nb < nblocks) {
                    const bf16x8 bh = cat4(tr_read(Th + ro + 32 * nb), tr_read(Th + ro + 32 * nb + 8 * LDR));
                    const bf16x8 bl = cat4(tr_read(Tl + ro + 32 * nb), tr_read(Tl + ro + 32 * nb + 8 * LDR));
                    O[nb] = mfma_bf16(pl, bh, O[nb]);
                    O[nb] = mfma_bf16(ph, bl, O[nb]);
                    O[nb] = mfma_bf16(ph, bh, O[nb]);
                }
            }
        }
    }
}
__device__ __forceinline__ void ptile_times_rows_pk(const bf16* Ph, const bf16* Pl, const bf16* Th, const bf16* Tl, int nblocks, int nks,
                                                    int lane, int wave, f32x16 (&O)[5], int nb0) {
    const int r = lane & 31, hh = lane >> 5, q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks >= nks) continue;
        const bf16x8 ah = *(const bf16x8*)(Ph + (32 * wave + r) * LDP + 16 * ks + 8 * hh);
        const bf16x8 al = *(const bf16x8*)(Pl + (32 * wave + r) * LDP + 16 * ks + 8 * hh);
        const int ro = (16 * ks + 8 * hh + q4) * LDR + 16 * g1 + 4 * p4;
#pragma unroll
        for (int nb = 0; nb < 5; ++nb) {
            if (nb >= nb0 && nb < nblocks) {
                const bf16x8 bh = cat4(tr_read(Th + ro + 32 * nb), tr_read(Th + ro + 32 * nb + 4 * LDR));
                const bf16x8 bl = cat4(tr_read(Tl + ro + 32 * nb), tr_read(Tl + ro + 32 * nb + 4 * LDR));
                O[nb] = mfma_bf16(al, bh, O[nb]);
                O[nb] = mfma_bf16(ah, bl, O[nb]);
                O[nb] = mfma_bf16(ah, bh, O[nb]);
            }
        }
    }
}
__device__ __forceinline__ void store_rows_pk(float* __restrict__ dst, const f32x16 (&O)[5], int wave, int lane, int rows, int H, int nb0,
                                              int nb1) {
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
        const int c = 32 * nb + r;
        if (c >= H || nb < nb0 || nb >= nb1) continue;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = 32 * wave + acc_row(j, hh);
            if (row < rows) dst[(size_t)row * H + c] = O[nb][j];
        }
    }
}

// The four-wave form of k_attn_x3_bwd (attn_x3.hip) on a tile: wave (w2 = wave & 1, half = wave >> 1) owns query / key rows 32 w2..
// and the channel blocks {0,1,2} (half 0) or {3,4} (half 1) of dQ / dV / dK.  A query's keys are [first row of its session, q].
__global__ __launch_bounds__(256) void k_attnp_bwd(AttnPkArgs a, AderSeqPack pk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* Th = (bf16*)smem_raw;
    bf16* Tl = Th + TR * LDR;
    bf16* Pdh = Tl + TR * LDR;                              // P_drop^T [key][query] hi/lo, dS^T hi/lo : [64][LDP] each
    bf16* Pdl = Pdh + TR * LDP;
    bf16* dSh = Pdl + TR * LDP;
    bf16* dSl = dSh + TR * LDP;
    float* km_l = (float*)(dSl + TR * LDP);
    int* tp_l = (int*)(km_l + TR);
    const int tile = blockIdx.x;
    if (tile >= pk.hdr[0]) return;
    const int nrows = pk.tile_rows[tile];
    const size_t prow0 = (size_t)tile * TR;
    // tiles of at most 32 rows (nearly all of them): there is ONE query / key block, so the four waves split the five channel blocks
    // {0,1} {2} {3} {4} instead of two of them idling, and the 64-row LDS tile holds TWO operands at once (V + K, then dO + Q): two
    // staging rounds and four workgroup barriers instead of four and eight
    const bool small = nrows <= 32;
    const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6;
    const int wave = small ? 0 : (part & 1), half = part >> 1;
    const int r = lane & 31, hh = lane >> 5;
    const int nb0 = small ? (part == 0 ? 0 : part + 1) : (half ? 3 : 0);       // this wave's channel blocks of dQ / dV / dK
    const int nb1 = small ? (part == 0 ? 2 : part + 2) : (half ? 5 : 3);
    const bool writer = small ? part == 0 : half == 0;                         // who writes the P_drop / dS tiles
    const int H = a.H;
    const size_t base = prow0 * H;
    const int ksteps = (H + 15) >> 4, nblocks = (H + 31) >> 5;
    const int nkb = nrows > 32 ? 2 : 1;                       // 32-row blocks that hold rows
    const int q = 32 * wave + r;
    const bool qok = q < nrows;
    for (int i = tid; i < (2 * TR * LDR + 4 * TR * LDP) / 2; i += 256) ((uint32_t*)Th)[i] = 0u;
    if (tid < TR) {
        km_l[tid] = (tid < nrows) ? a.kmask[prow0 + tid] : 0.0f;
        tp_l[tid] = (tid < nrows) ? ((pk.info[prow0 + tid] >> 8) & 63) : 0;
    }
    const int seg0 = qok ? (pk.info[prow0 + q] & 63) : q;
    const uint32_t dq = qok ? pk.gpos[prow0 + q] * (uint32_t)a.T : 0u;
    f32x16 X[2];                                            // dP^T, then dS^T (keys on rows, this wave's queries on lanes)
    {
        bf16x8 gh[10], gl[10];
        row_frags_pk(a.dO + base + (size_t)(qok ? q : 0) * H, qok, H, hh, gh, gl);              // dO rows
        __syncthreads();
        stage_split_pk<256>(Th, Tl, a.V + base, nrows, H, tid);
        if (small) stage_split_pk<256>(Th + 32 * LDR, Tl + 32 * LDR, a.K + base, nrows, H, tid);   // K behind V: rows 32..
        __syncthreads();
        rows_times_frags_pk(Th, Tl, gh, gl, ksteps, nkb, r, hh, X);                              // dP_drop^T = V . dO^T
    }
    // softmax backward for query q
    const float qm = qok ? a.qmask[prow0 + q] : 0.0f;
    const float* PTt = a.PT + prow0 * TR;
    f32x16 Pv[2];
    float dot = 0.0f;
    // the saved probabilities of this lane: UNCONDITIONAL loads (element 0 of the tile where the entry does not exist) issued together
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            const bool in = qok && key >= seg0 && key <= q;
            Pv[kb][j] = PTt[in ? key * TR + q : 0];
        }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            const bool in = qok && key >= seg0 && key <= q;
            float p = 0.0f, dp = 0.0f, pd = 0.0f;
            if (in) {
                p = Pv[kb][j];
                float f = qm;
                if (a.thr != 0) f = ((lowbias32((dq + (uint32_t)tp_l[key]) ^ a.key) >> 8) >= a.thr) ? f * a.scale : 0.0f;
                dp = X[kb][j] * f;
                pd = p * f;
                dot += dp * p;
            }
            Pv[kb][j] = p;
            X[kb][j] = dp;
            const bf16 ph = (bf16)pd;
            if (writer) {
                Pdh[key * LDP + q] = ph;
                Pdl[key * LDP + q] = (bf16)(pd - (float)ph);
            }
        }
    dot += __shfl_xor(dot, 32, 64);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = 32 * kb + acc_row(j, hh);
            const bool in = qok && key >= seg0 && key <= q;
            float ds = 0.0f;
            if (in && km_l[key] != 0.0f) ds = (Pv[kb][j] * (X[kb][j] - dot)) / a.sqrt_dh;
            X[kb][j] = ds;
            const bf16 sh_ = (bf16)ds;
            if (writer) {
                dSh[key * LDP + q] = sh_;
                dSl[key * LDP + q] = (bf16)(ds - (float)sh_);
            }
        }
    f32x16 O[5];
    const int nks = nrows > 32 ? 4 : 2;                     // 16-query steps that hold rows
#define ZERO_O() _Pragma("unroll") for (int nb = 0; nb < 5; ++nb) _Pragma("unroll") for (int j = 0; j < 16; ++j) O[nb][j] = 0.0f;
    if (small) {
        // dQ[q][c] = sum_key dS^T[key][q] K[key][c]: K is already staged (rows 32..)
        ZERO_O()
        acc_times_rows_pk(X, Th + 32 * LDR, Tl + 32 * LDR, min(nblocks, nb1), 1, lane, O, nb0);
        store_rows_pk(a.dQ + base, O, 0, lane, nrows, H, nb0, nb1);
        __syncthreads();                                   // V / K consumed; the P_drop / dS tiles are written
        stage_split_pk<256>(Th, Tl, a.dO + base, nrows, H, tid);
        stage_split_pk<256>(Th + 32 * LDR, Tl + 32 * LDR, a.Q + base, nrows, H, tid);
        __syncthreads();
        // dV[key][c] = sum_q P_drop^T[key][q] dO[q][c];  dK[key][c] = sum_q dS^T[key][q] Q[q][c]
        ZERO_O()
        ptile_times_rows_pk(Pdh, Pdl, Th, Tl, min(nblocks, nb1), nks, lane, 0, O, nb0);
        store_rows_pk(a.dV + base, O, 0, lane, nrows, H, nb0, nb1);
        ZERO_O()
        ptile_times_rows_pk(dSh, dSl, Th + 32 * LDR, Tl + 32 * LDR, min(nblocks, nb1), nks, lane, 0, O, nb0);
        store_rows_pk(a.dK + base, O, 0, lane, nrows, H, nb0, nb1);
        return;
    }
    // dQ[q][c] = sum_key dS^T[key][q] K[key][c]   (accumulator operand)
    __syncthreads();
    stage_split_pk<256>(Th, Tl, a.K + base, nrows, H, tid);
    __syncthreads();
    ZERO_O()
    if (wave < nkb) acc_times_rows_pk(X, Th, Tl, min(nblocks, nb1), nkb, lane, O, nb0);
    store_rows_pk(a.dQ + base, O, wave, lane, nrows, H, nb0, nb1);
    // dV[key][c] = sum_q P_drop^T[key][q] dO[q][c]   (wave = key block)
    __syncthreads();
    stage_split_pk<256>(Th, Tl, a.dO + base, nrows, H, tid);
    __syncthreads();
    ZERO_O()
    if (wave < nkb) ptile_times_rows_pk(Pdh, Pdl, Th, Tl, min(nblocks, nb1), nks, lane, wave, O, nb0);
    store_rows_pk(a.dV + base, O, wave, lane, nrows, H, nb0, nb1);
    // dK[key][c] = sum_q dS^T[key][q] Q[q][c]
    __syncthreads();
    stage_split_pk<256>(Th, Tl, a.Q + base, nrows, H, tid);
    __syncthreads();
    ZERO_O()
    if (wave < nkb) ptile_times_rows_pk(dSh, dSl, Th, Tl, min(nblocks, nb1), nks, lane, wave, O, nb0);
    store_rows_pk(a.dK + base, O, wave, lane, nrows, H, nb0, nb1);
#undef ZERO_O
}

// ------------------------------------------------------------------------------------------------ pruned block: one query per session
struct AttnLastPkArgs {
    const float *Ql, *K, *V, *dO;          // Ql, dO compact [B,H]; K, V tile order
    const float *P;                         // compact [B,T]: probabilities of the last query over the session's positions
    const float *kmask, *qmask;            // kmask tile order, qmask compact [B]
    float *dQl, *dK, *dV;
    int T, H;
    float sqrt_dh;
    uint32_t key, thr;
    float scale;
};
__device__ __forceinline__ float sum16_pk(float v) { return row16_sum(v); }

// k_attn_last_bwd (attn.hip) with the session's keys at packed rows [srow0, srow0 + slen): one workgroup per session
__global__ __launch_bounds__(256) void k_attnp_last_bwd(AttnLastPkArgs a, AderSeqPack pk) {
    __shared__ float acc_l[TR], pd_l[TR], ds_l[TR];
    __shared__ float part_l[16][160];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, rsub = lane >> 4, grp = 4 * wave + rsub;        // row group 0..15: rows grp, grp + 16, ...
    const int b = blockIdx.x;
    const int T = a.T, H = a.H;
    const int len = pk.slen[b];
    const size_t base = (size_t)pk.srow0[b] * H;                                // first packed row of the session
    float g[10], q[10], v[4][10], kk[4][10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int c = sub + 16 * i;
        g[i] = (c < H) ? a.dO[(size_t)b * H + c] : 0.0f;
        q[i] = (c < H) ? a.Ql[(size_t)b * H + c] : 0.0f;
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int t = grp + 16 * ps;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = sub + 16 * i;
            const bool ok = t < len && c < H;
            v[ps][i] = ok ? a.V[base + (size_t)t * H + c] : 0.0f;
            kk[ps][i] = ok ? a.K[base + (size_t)t * H + c] : 0.0f;
        }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 10; ++i) s = fmaf(g[i], v[ps][i], s);
        s = sum16_pk(s);
        const int t = grp + 16 * ps;
        if (sub == 0 && t < TR) acc_l[t] = s;
    }
    __syncthreads();
    if (tid < 64) {
        float p = 0.0f, dp = 0.0f, pd = 0.0f;
        if (lane < len) {
            const int tp = T - len + lane;                                      // position of the session's lane-th row
            p = a.P[(size_t)b * T + tp];
            float f = a.qmask[b];
            if (a.thr != 0) {
                const uint32_t didx = pk.gpos[pk.srow0[b] + len - 1] * (uint32_t)T + (uint32_t)tp;
                f = ((lowbias32(didx ^ a.key) >> 8) >= a.thr) ? f * a.scale : 0.0f;
            }
            dp = acc_l[lane] * f;
            pd = p * f;
        }
        const float dot = wave_sum(dp * p);
        pd_l[lane] = pd;
        ds_l[lane] = (lane < len && a.kmask[pk.srow0[b] + lane] != 0.0f) ? (p * (dp - dot)) / a.sqrt_dh : 0.0f;
    }
    __syncthreads();
    float dq[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) dq[i] = 0.0f;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int t = grp + 16 * ps;
        if (t < len) {
            const float pd = pd_l[t], ds = ds_l[t];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int c = sub + 16 * i;
                if (c < H) {
                    a.dV[base + (size_t)t * H + c] = pd * g[i];
                    a.dK[base + (size_t)t * H + c] = ds * q[i];
                }
                dq[i] = fmaf(ds, kk[ps][i], dq[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) part_l[grp][sub + 16 * i] = dq[i];
    __syncthreads();
    if (tid < H) {
        float s = 0.0f;
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) s += part_l[gi][tid];       // fixed order: bit-reproducible
        a.dQl[(size_t)b * H + tid] = s;
    }
}

// dpos[t][c] = sum over the sessions that HAVE position t (t >= T - slen[b]) of g[b*T + t][c]: k_pos_grad (rowwise.hip) for the
// session-indexed gradient rows of a packed step, whose padding rows are never written.  8 b-slices, fixed order.
__global__ __launch_bounds__(256) void k_pos_grad_pk(const float* __restrict__ g, const int* __restrict__ slen, float* __restrict__ dpos,
                                                     int B, int T, int H) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    const int TH = T * H;
    float acc = 0.0f;
    if (i < TH) {
        const int t = i / H;
        const int per = (B + 7) / 8;
        const int b0 = sl * per, b1 = min(B, b0 + per);
        // (unconditional loads -- the rows exist, stale or not -- and a select: a load under the branch is waited for inside it,
        //  one memory round trip per session)
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            float v[4];
            int ln[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { v[u] = g[(size_t)(b + u) * TH + i]; ln[u] = slen[b + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += (t >= T - ln[u]) ? v[u] : 0.0f;
        }
        for (; b < b1; ++b) { const float v = g[(size_t)b * TH + i]; acc += (t >= T - slen[b]) ? v : 0.0f; }
    }
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < TH) {
        float a = red[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) a += red[k][o];
        dpos[i] = a;
    }
}

static const size_t kSeqpBwdLds = (size_t)3 * RSZ * sizeof(bf16) + (size_t)RED_FLOATS * sizeof(float) + 4 * TR * sizeof(int);
static const size_t kAttnpBwdLds = (size_t)(2 * TR * LDR + 4 * TR * LDP) * sizeof(bf16) + 2 * TR * sizeof(float);

static int check_dims(int B, int T, int H) {
    if (B <= 0) return 1;
    if (T < 1 || T > TR || H < 2 || H > 150 || (H & 1) || B > 4096) return -2;
    return 0;
}
template <class K> static int set_lds(K kernel, size_t bytes, bool& done) {
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    return 0;
}

extern "C" {

int ader_seqp_bwd_ffn(const AderSeqBwdFfn* desc, const AderSeqPack* pack, int max_tiles, void* stream) {
    const AderSeqBwdFfn& a = *desc;
    const int rc = check_dims(a.B, a.T, a.H);
    if (rc) return rc < 0 ? rc : 0;
    if (!pack) return -2;
    if (max_tiles <= 0 || max_tiles > a.B) max_tiles = a.B;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (int e = set_lds(k_seqp_bwd_ffn, kSeqpBwdLds, attr_set)) return e;
    hipLaunchKernelGGL(k_seqp_bwd_ffn, dim3(max_tiles), dim3(640), kSeqpBwdLds, (hipStream_t)stream, a, *pack);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_seqp_bwd_qkv(const AderSeqBwdQkv* desc, const AderSeqPack* pack, int max_tiles, void* stream) {
    const AderSeqBwdQkv& a = *desc;
    const int rc = check_dims(a.B, a.T, a.H);
    if (rc) return rc < 0 ? rc : 0;
    if (!pack) return -2;
    if (max_tiles <= 0 || max_tiles > a.B) max_tiles = a.B;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (int e = set_lds(k_seqp_bwd_qkv, kSeqpBwdLds, attr_set)) return e;
    hipLaunchKernelGGL(k_seqp_bwd_qkv, dim3(max_tiles), dim3(640), kSeqpBwdLds, (hipStream_t)stream, a, *pack);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_attnp_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* PT, const float* kmask, const float* qmask,
                   float* dQ, float* dK, float* dV, int B, int T, int H, const AderDrop* drop, const AderSeqPack* pack, int max_tiles,
                   void* stream) {
    const int rc = check_dims(B, T, H);
    if (rc) return rc < 0 ? rc : 0;
    if (!pack) return -2;
    if (max_tiles <= 0 || max_tiles > B) max_tiles = B;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (int e = set_lds(k_attnp_bwd, kAttnpBwdLds, attr_set)) return e;
    AttnPkArgs a;
    a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.kmask = kmask; a.qmask = qmask; a.PT = PT; a.dQ = dQ; a.dK = dK; a.dV = dV;
    a.H = H; a.T = T; a.sqrt_dh = sqrtf((float)H);
    a.key = drop ? drop->key : 0u; a.thr = drop ? drop->thr : 0u; a.scale = drop ? drop->scale : 1.0f;
    hipLaunchKernelGGL(k_attnp_bwd, dim3(max_tiles), dim3(256), kAttnpBwdLds, (hipStream_t)stream, a, *pack);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_attnp_last_bwd(const float* dO_last, const float* Q_last, const float* K, const float* V, const float* P_last, const float* kmask,
                        const float* qmask_last, float* dQ_last, float* dK, float* dV, int B, int T, int H, const AderDrop* drop,
                        const AderSeqPack* pack, void* stream) {
    const int rc = check_dims(B, T, H);
    if (rc) return rc < 0 ? rc : 0;
    if (!pack) return -2;
    AttnLastPkArgs a;
    a.Ql = Q_last; a.K = K; a.V = V; a.dO = dO_last; a.P = P_last; a.kmask = kmask; a.qmask = qmask_last;
    a.dQl = dQ_last; a.dK = dK; a.dV = dV; a.T = T; a.H = H; a.sqrt_dh = sqrtf((float)H);
    a.key = drop ? drop->key : 0u; a.thr = drop ? drop->thr : 0u; a.scale = drop ? drop->scale : 1.0f;
    hipLaunchKernelGGL(k_attnp_last_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, a, *pack);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_pos_grad_packed(const float* dx, const int* slen, float* dpos, int B, int T, int H, void* stream) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_pos_grad_pk, dim3((T * H + 31) / 32), dim3(256), 0, (hipStream_t)stream, dx, slen, dpos, B, T, H);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
