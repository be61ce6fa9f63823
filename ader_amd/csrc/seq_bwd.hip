// Session-tiled backward chains of one SASRec block (tf.gradients of the graph built at ADER.py:62-81): the row-local
// kernels on either side of the attention backward, each as ONE launch of B workgroups that keep the session's rows in LDS.
//
//   ader_seq_bwd_ffn : dx2 -> [mask, dropout-grad] -> .W2^T -> [relu/dropout-grad] -> .W1^T (+ residual) -> LayerNorm2 backward
//                      replaces mask_dropgrad, gemm_x3<RELUDROPGRAD>, gemm_x3<ADD>, ln_bwd (+ its slab pass) of a block
//   ader_seq_bwd_qkv : dQ.Wq^T + dx1 -> LayerNorm1 backward, + dK.Wk^T + dV.Wv^T [-> embedding-prologue backward for block 0]
//                      replaces gemm_x3<ADD> x3, ln_bwd, add_rows (+ embed_bwd_rows)
// They write the same tensors as the per-op kernels (the weight-gradient operands dh2/da stay in memory for the batched
// A^T.G launch), in the same layouts -- compact [B,..] tensors of position T-1 for a pruned (last) block -- and per-session
// partial sums of the LayerNorm gamma/beta gradients ([B][2][H], reduced in a fixed order by ader_reduce_slabs).
// Same arithmetic as the per-op path: bf16 hi/lo split operands, 3 MFMAs per product, fp32 accumulation and row math.
// heads-agnostic (attention is not in these kernels).  T <= 64, H even <= 150.  gfx950 only.
#include "seq_common.h"

#define RED_FLOATS (10 * 2 * HP)          // per-wave gamma/beta partials

// Row-layout helpers: 16 lanes per row, lane (rsub = lane >> 4, sub = lane & 15) owns columns sub + 16 i of row
// 40 * pass + 4 * wave + rsub.

// dgamma/dbeta partials of the waves -> slab[b][2][H] (fixed order: lanes, then waves 0..9)
__device__ __forceinline__ void flush_ln_partials(float (&dg)[10], float (&db)[10], float* red_l, float* __restrict__ slab, int b, int H,
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
        if (c < H) slab[((size_t)b * 2 + which) * H + c] = s;
    }
}

// LayerNorm backward of the session's rows (reference LN: modules.py:44-48).  dy rows in Xf (fp32 LDS), x rows in memory.
//   dx = (1/sd) * (dxh - mean(dxh) - xh*mean(dxh*xh)), dxh = dy*gamma, xh = (x-mean)/sd;  dgamma += dy*xh, dbeta += dy.
// keep_row(t): rows that exist (t < T, and only T-1 for a pruned block).  dx -> Xf in place and, if odx, to memory.
template <bool STORE>
__device__ __forceinline__ void ln_bwd_rows(float* Xf, const Out& ox, const Out& omean, const Out& ostd, const float (&gam)[10],
                                            const Out& odx, int T, int H, bool pruned, int tv0, int wave, int lane, float (&dg)[10],
                                            float (&db)[10]) {
    const int sub = lane & 15, rsub = lane >> 4;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const int t = 40 * pass + 4 * wave + rsub;
        if (t >= TR) continue;
        const bool valid = t < T && (!pruned || t == T - 1);
        const uint32_t bo = (uint32_t)(t * H + sub) * 4u;
        if (t < tv0) {      // leading padding: the forward wrote no x / mean / sd for these rows; their gradient is exactly zero
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int c = sub + 16 * i;
                Xf[t * XS + c] = 0.0f;
                if (STORE) bstore(odx, (c < H) ? bo + 64u * i : OOB, 0.0f);
            }
            continue;
        }
        float xv[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) xv[i] = bload(ox, (sub + 16 * i < H) ? bo + 64u * i : OOB);
        const float mean = bload(omean, (uint32_t)t * 4u);
        float sd = bload(ostd, (uint32_t)t * 4u);
        if (!valid) sd = 1.0f;
        const float rsd = 1.0f / sd;                        // one reciprocal per row instead of 20 divisions
        float xh[10], dxh[10], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = sub + 16 * i;
            const float g = (valid && c < H) ? Xf[t * XS + c] : 0.0f;
            xh[i] = (valid && c < H) ? (xv[i] - mean) * rsd : 0.0f;
            dxh[i] = g * gam[i];
            s1 += dxh[i];
            s2 += dxh[i] * xh[i];
            dg[i] += g * xh[i];
            db[i] += g;
        }
        s1 = row16_sum(s1) / (float)H;
        s2 = row16_sum(s2) / (float)H;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = sub + 16 * i;
            const float o = (valid && c < H) ? (dxh[i] - s1 - xh[i] * s2) * rsd : 0.0f;
            Xf[t * XS + c] = o;
            if (STORE) bstore(odx, (c < H) ? bo + 64u * i : OOB, o);
        }
    }
}

// rows of a [.,H] fp32 tensor (through its descriptor: absent rows read as 0) -> hi/lo tile
__device__ __forceinline__ void stage_rows(const Out& src, bf16* R, int H, int wave, int lane) {
    const int sub = lane & 15, rsub = lane >> 4;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const int t = 40 * pass + 4 * wave + rsub;
        if (t < TR) {
            const uint32_t bo = (uint32_t)(t * H + sub) * 4u;
            float v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) v[i] = bload(src, (sub + 16 * i < H) ? bo + 64u * i : OOB);
            bf16* Tp = R + t * LDR + sub;
#pragma unroll
            for (int i = 0; i < 10; ++i) put_split(Tp, Tp + TR * LDR, 16 * i, v[i]);
        }
    }
}

__global__ __launch_bounds__(640) void k_seq_bwd_ffn(AderSeqBwdFfn a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R0 = (bf16*)smem_raw;
    bf16* R1 = R0 + RSZ;
    float* Xf = (float*)(R1 + RSZ);
    float* red_l = (float*)((bf16*)Xf + RSZ);
    int* sq_l = (int*)(red_l + RED_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = wave % 5, mh = wave / 5;
    const int b = blockIdx.x, T = a.T, H = a.H;
    const bool pruned = a.pruned != 0;
    const uint32_t H4 = (uint32_t)H * 4u;
    const uint32_t didx_row0 = (uint32_t)b * (uint32_t)T * (uint32_t)H;
    bf16x8 bh[10], bl[10];
    {
        const int r = lane & 31, hh = lane >> 5;
        load_bfrags((const bf16*)a.w2 + 2 * WSZ, nb, r, hh, bh, bl);        // W2 planes [n][k] = W2[n][k]: A . W2^T
    }
    if (tid < TR) sq_l[tid] = (tid < T) ? a.seq[(size_t)b * T + tid] : 0;
    float gam[10];
    load10(a.ln2_g, H, lane & 15, gam);
    lds_barrier();
    // leading padding (see seq_fwd.hip): rows [0, tv0) carry exactly zero gradient and the forward saved nothing for them -- they
    // are not computed, their gradient rows are written as zeros (the weight-gradient products and the attention backward read
    // every row)
    int tv0;
    {
        const unsigned long long nz = __ballot(lane < T && sq_l[lane] != 0);
        tv0 = nz ? (int)__ffsll((long long)nz) - 1 : T;
    }
    const bool skipw = mh == 0 && tv0 >= 32;
    // ---- g = dx2 * (seq != 0) -> Xf;  dh2 = g * keep * scale (modules.py:262-266, ADER.py:80) -> memory, hi/lo -> R0
    {
        const Out odx = make_out(a.dx2, b, T, H, pruned), odh = make_out(a.dh2, b, T, H, pruned);
        const SDrop d2 = sdrop_of(a.d_ffn2, didx_row0);
        const int sub = lane & 15, rsub = lane >> 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int t = 40 * pass + 4 * wave + rsub;
            if (t < tv0) {
                const uint32_t bo = (uint32_t)(t * H + sub) * 4u;
                bf16* Tp = R0 + t * LDR + sub;
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int c = sub + 16 * i;
                    Xf[t * XS + c] = 0.0f;
                    Tp[16 * i] = (bf16)0.0f; Tp[TR * LDR + 16 * i] = (bf16)0.0f;
                    bstore(odh, (c < H) ? bo + 64u * i : OOB, 0.0f);
                }
            } else if (t < TR) {
                const uint32_t bo = (uint32_t)(t * H + sub) * 4u;
                float v[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) v[i] = bload(odx, (sub + 16 * i < H) ? bo + 64u * i : OOB);
                const bool live = sq_l[t] != 0;
                bf16* Tp = R0 + t * LDR + sub;
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const int c = sub + 16 * i;
                    const float g = live ? v[i] : 0.0f;
                    Xf[t * XS + c] = g;
                    const float dh = sdrop_apply1(d2, didx_row0 + d2.off + (uint32_t)(t * H + c), g);
                    put_split(Tp, Tp + TR * LDR, 16 * i, dh);
                    bstore(odh, (c < H) ? bo + 64u * i : OOB, dh);
                }
            }
        }
    }
    lds_barrier();
    // ---- da = (dh2 . W2^T) * relu/dropout-grad (modules.py:254-257) -> memory, hi/lo -> R1
    {
        PHASE_IDS;
        const Out oh = make_out(a.h1d, b, T, H, pruned), oda = make_out(a.da, b, T, H, pruned);
        const int t0 = 32 * mh + 4 * hh;
        const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
        float h1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) h1[j] = bload(oh, boff0 + ROWJ(j) * H4);
        f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
        load_bfrags((const bf16*)a.w1 + 2 * WSZ, nb, r, hh, bh, bl);
        bf16* Th = R1 + t0 * LDR + n;
        const float sc1 = a.d_ffn1.scale;
        if (skipw) {
#pragma unroll
            for (int j = 0; j < 16; ++j) bstore(oda, boff0 + ROWJ(j) * H4, 0.0f);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = (h1[j] != 0.0f) ? acc[j] * sc1 : 0.0f;
                put_split(Th, Th + TR * LDR, ROWJ(j) * LDR, v);
                bstore(oda, boff0 + ROWJ(j) * H4, v);
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
    // ---- LayerNorm2 backward -> dx1 (memory) + gamma/beta partials of the session
    float dg[10], db[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { dg[i] = 0.0f; db[i] = 0.0f; }
    {
        const Out ox = make_out(a.x1, b, T, H, pruned), om = make_out(a.mean2, b, T, 1, pruned), os = make_out(a.std2, b, T, 1, pruned);
        const Out odx1 = make_out(a.dx1, b, T, H, pruned);
        ln_bwd_rows<true>(Xf, ox, om, os, gam, odx1, T, H, pruned, tv0, wave, lane, dg, db);
    }
    flush_ln_partials(dg, db, red_l, a.slab, b, H, tid);
}

__global__ __launch_bounds__(640) void k_seq_bwd_qkv(AderSeqBwdQkv a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* R0 = (bf16*)smem_raw;
    bf16* R1 = R0 + RSZ;
    float* Xf = (float*)(R1 + RSZ);
    float* red_l = (float*)((bf16*)Xf + RSZ);
    int* sq_l = (int*)(red_l + RED_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = wave % 5, mh = wave / 5;
    const int b = blockIdx.x, T = a.T, H = a.H;
    const bool pruned = a.pruned != 0;
    const uint32_t H4 = (uint32_t)H * 4u;
    bf16x8 bh[10], bl[10];
    {
        const int r = lane & 31, hh = lane >> 5;
        load_bfrags((const bf16*)a.wq + 2 * WSZ, nb, r, hh, bh, bl);
    }
    if (tid < TR) sq_l[tid] = (a.emb_bwd && tid < T) ? a.seq[(size_t)b * T + tid] : 1;
    int tv0;                                    // leading padding rows (see k_seq_bwd_ffn)
    {
        const unsigned long long nz = __ballot(lane < T && a.seq[(size_t)b * T + (lane < T ? lane : 0)] != 0);
        tv0 = nz ? (int)__ffsll((long long)nz) - 1 : T;
    }
    const bool skipw = mh == 0 && tv0 >= 32;
    float gam[10];
    load10(a.ln1_g, H, lane & 15, gam);
    // ---- dQ rows -> R0 (a pruned block has the row of position T-1 only), dK rows -> R1
    {
        const Out oq = make_out(a.dQ, b, T, H, pruned);
        stage_rows(oq, R0, H, wave, lane);
        const Out ok = make_out(a.dK, b, T, H, false);
        stage_rows(ok, R1, H, wave, lane);
    }
    lds_barrier();
    // ---- dqin = dQ . Wq^T + dx1 -> Xf
    {
        PHASE_IDS;
        const Out ox1 = make_out(a.dx1, b, T, H, pruned);
        const int t0 = 32 * mh + 4 * hh;
        const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
        float res[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) res[j] = bload(ox1, boff0 + ROWJ(j) * H4);
        f32x16 acc = tile_mma(R0, mh, r, hh, bh, bl, skipw);
        load_bfrags((const bf16*)a.wk + 2 * WSZ, nb, r, hh, bh, bl);
        float* Xp = Xf + t0 * XS + n;
        if (!skipw) {           // (an all-padding wave: its Xf rows are cleared by the LayerNorm backward below)
#pragma unroll
            for (int j = 0; j < 16; ++j) Xp[ROWJ(j) * XS] = acc[j] + res[j];
        }
    }
    lds_barrier();
    // ---- dV rows -> R0 (dQ is consumed); LayerNorm1 backward in place in Xf + gamma/beta partials
    float dg[10], db[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { dg[i] = 0.0f; db[i] = 0.0f; }
    {
        const Out ov = make_out(a.dV, b, T, H, false);
        stage_rows(ov, R0, H, wave, lane);
        // x rows of the block input are [B*T,H] also for a pruned block; its statistics are compact
        Out ox = make_out(a.x, b, T, H, false);
        const Out om = make_out(a.mean1, b, T, 1, pruned), os = make_out(a.std1, b, T, 1, pruned);
        ln_bwd_rows<false>(Xf, ox, om, os, gam, ox, T, H, pruned, tv0, wave, lane, dg, db);
    }
    lds_barrier();
    // ---- dx = LN1-backward + dK . Wk^T + dV . Wv^T  [block 0: * (seq != 0) * keep * scale of the embedding prologue]
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
        const Out odx = make_out(a.dx, b, T, H, false);
        const SDrop d0 = sdrop_of(a.d_emb, (uint32_t)b * (uint32_t)T * (uint32_t)H);
        const int t0 = 32 * mh + 4 * hh;
        const uint32_t boff0 = (n < H) ? (uint32_t)(t0 * H + n) * 4u : OOB;
        const uint32_t didx0 = (uint32_t)b * (uint32_t)T * (uint32_t)H + d0.off + (uint32_t)(t0 * H + n);
        const float* Xp = Xf + t0 * XS + n;
        if (skipw) {
#pragma unroll
            for (int j = 0; j < 16; ++j) bstore(odx, boff0 + ROWJ(j) * H4, 0.0f);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v = Xp[ROWJ(j) * XS] + acc[j];
                if (a.emb_bwd) {
                    v = sdrop_apply1(d0, didx0 + ROWJ(j) * (uint32_t)H, v);
                    v = (sq_l[t0 + ROWJ(j)] != 0) ? v : 0.0f;
                }
                bstore(odx, boff0 + ROWJ(j) * H4, v);
            }
        }
    }
    flush_ln_partials(dg, db, red_l, a.slab, b, H, tid);
}

static const size_t kSeqBwdLds = (size_t)3 * RSZ * sizeof(bf16) + (size_t)RED_FLOATS * sizeof(float) + TR * sizeof(int);

static int check_dims(int B, int T, int H) {
    if (B <= 0) return 1;
    if (T < 1 || T > TR || H < 2 || H > 150 || (H & 1)) return -2;
    return 0;
}

extern "C" {

int ader_seq_bwd_ffn(const AderSeqBwdFfn* desc, void* stream) {
    const AderSeqBwdFfn& a = *desc;
    const int rc = check_dims(a.B, a.T, a.H);
    if (rc) return rc < 0 ? rc : 0;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_seq_bwd_ffn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSeqBwdLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_seq_bwd_ffn, dim3(a.B), dim3(640), kSeqBwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

int ader_seq_bwd_qkv(const AderSeqBwdQkv* desc, void* stream) {
    const AderSeqBwdQkv& a = *desc;
    const int rc = check_dims(a.B, a.T, a.H);
    if (rc) return rc < 0 ? rc : 0;
    static bool attr_set_dev[ADER_MAX_DEV] = {};
    bool& attr_set = attr_set_dev[ader_cur_dev()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_seq_bwd_qkv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSeqBwdLds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_seq_bwd_qkv, dim3(a.B), dim3(640), kSeqBwdLds, (hipStream_t)stream, a);
    HIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
