/* ader_hip.h -- C ABI of libader_hip.so: the MI355X (gfx950) kernels of the ADER / SASRec training hot path.
 *
 * The reference (doublemul/ADER) is pure Python/TensorFlow and has NO native interface: the boundary these
 * entry points replace is the set of TF ops executed by `sess.run` at main.py:233-256 (train step),
 * util.py:323 -> ADER.py:140-150 (ranking) and util.py:452-455 + util.py:401-434 (exemplar selection).
 * Each function below names the reference op site it replaces.
 *
 * Conventions: plain device pointers (float32 / int32 unless noted), sizes as ints, `stream` is a hipStream_t
 * passed as void* (NULL = default stream).  Launchers only enqueue work (no sync, no allocation: graph-capturable);
 * they return 0 on success, a hipError_t value on a HIP error, or a negative code for unsupported shapes
 * (-2: dimension out of range, -3: bad enum).  Inputs are borrowed for the duration of the enqueued work.
 * Limits of this build: H (hidden_units) <= 159, maxlen T <= 64; padded batch rows Bp % 64 == 0 and Bp <= 1024 for the exact-f32
 * logit kernels (ader_logits_*: per-row state in LDS); the flash kernels (ader_lbf_*, ader_lx3_*, ader_tab_*) take Bp % 128 == 0 of
 * any size (the engine allows 4096 rows per step).
 */
#ifndef ADER_HIP_H
#define ADER_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Dropout everywhere (tf.layers.dropout sites ADER.py:55, modules.py:214,257,262; TF's RNG stream is not reproducible, the
 * counter spec is the build's own and is restated in oracle/ader_ref_cpu.py):
 *   keep(idx) = (lowbias32((idx + offset(idx)) ^ key) >> 8) >= thr,  value * scale when kept;  thr == 0 (or a NULL
 *   descriptor) disables it.  idx is the element's LOCAL index at the site (row * elements_per_row + ...), offset(idx) makes it
 *   global: `base` for idx < split, `base2` from `split` on -- a data-parallel rank holds a slice of the train rows followed by
 *   a slice of the exemplar rows (main.py:229), two row segments with different global positions; split = 0xFFFFFFFF: one. */
typedef struct { unsigned key, thr; float scale; unsigned base, split, base2; } AderDrop;

/* ---- embedding prologue: modules.py:118-130 + ADER.py:41-60 ------------------------------------------- */
int ader_embed_fwd(const int* seq, const float* emb, const float* pos, float* x, int rows, int T, int H, int V,
                   const AderDrop* drop, int* status, void* stream);
/* backward of the prologue: dx [B*T,H] is overwritten with the masked / dropout-scaled per-position gradient rows (consumed by the
 * fused table update through an id-sorted list, or added into demb by ader_scatter_rows_ordered); dpos [T,H] is overwritten.  seq == NULL: dx already holds those rows
 * (ader_seq_bwd_qkv with emb_bwd) and only dpos is computed. */
int ader_embed_bwd_rows(const int* seq, float* dx, float* dpos, int B, int T, int H, int V, const AderDrop* drop, void* stream);
/* demb[ids[k]] += rows[rws[k]] * scale without atomics (the scatter of the gather's gradient, modules.py:127 differentiated, for rows
 * already masked / dropout-scaled by ader_embed_bwd_rows): (ids, rws, start) = the bucketed lists ader_sparse_lists builds from the ids
 * (nb buckets); every table row is summed in position order -- bit-identical on every data-parallel rank that holds the same gathered rows (modules.py:127 gradient) */
int ader_scatter_rows_ordered(const int* ids, const int* rws, const int* start, int nb, const float* rows, int H, int V, float scale,
                              float* demb, void* stream);

/* ---- LayerNorm: modules.py:23-50 (`normalize`) ---------------------------------------------------------- */
/* xnz/ynz (optional) = sign(|sum_c x|), sign(|sum_c y|): the key / query masks of modules.py:188,208. */
int ader_ln_fwd(const float* x, long x_row_stride, float* y, long y_row_stride, const float* gamma, const float* beta,
                float* mean_out, float* std_out, float* xnz, float* ynz, int rows, int H, void* stream);
int ader_ln_bwd_slabs(int rows);
/* dx = LN'(dy) (+ add); dgamma/dbeta overwritten.  slab: ader_ln_bwd_slabs(rows)*2*H floats of scratch. */
int ader_ln_bwd(const float* dy, long dy_row_stride, const float* x, long x_row_stride, const float* gamma,
                const float* mean_in, const float* std_in, const float* add, long add_row_stride, float* dx,
                long dx_row_stride, float* slab, float* dgamma, float* dbeta, int rows, int H, void* stream);

/* ---- dense layers: modules.py:172-174 (Q/K/V), modules.py:254-266 (FFN) -------------------------------- */
enum { ADER_EPI_BIAS = 0, ADER_EPI_BIAS_RELU_DROP = 1, ADER_EPI_BIAS_DROP_RES_MASK = 2, ADER_EPI_RELUDROPGRAD = 3,
       ADER_EPI_ADD = 4 };
/* C[M,H] = epilogue(A[M,H] . (trans_b ? W^T : W) + bias).  aux/seq per epilogue, see csrc/gemm.hip. */
/* row_mul/row_add: local row m is row m*row_mul+row_add of the full [B*T,H] tensor (dropout counter and seq mask of a row
 * subset, e.g. only position T-1 of every sequence: row_mul = T, row_add = T-1); 1, 0 for full tensors. */
int ader_gemm_rows(const float* A, const float* W, const float* bias, float* C, const float* aux, const int* seq, int M,
                   int H, int epilogue, int trans_b, int row_mul, int row_add, const AderDrop* drop, void* stream);
int ader_gemm_atb_slabs(int M);
/* dW[H,H] = A^T . G, db[H] = column sums of G (db may be NULL).  slab: ader_gemm_atb_slabs(M)*160*160 floats. */
int ader_gemm_atb(const float* A, const float* G, float* slab, float* dW, float* db, int M, int H, void* stream);
/* "bf16x3" variants on v_mfma_f32_32x32x16_bf16: operands split into bf16 hi+lo, 3 MFMAs per product, fp32 accumulate
 * (float32-grade accuracy, ~2^-16 relative per product).  Same semantics/epilogues as ader_gemm_rows / ader_gemm_atb.
 * ader_wprep builds, for nw [H,H] weights at theta+offs[i], 4 planes each of [160][168] bf16: W^T hi, W^T lo, W hi, W lo. */
size_t ader_wprep_elems(int nw);
int ader_wprep(const float* theta, const long* offs, int nw, int H, void* out, void* stream);
int ader_gemm_x3(const float* A, const void* wplanes, const float* bias, float* C, const float* aux, const int* seq, int M,
                 int H, int epilogue, int trans_b, int row_mul, int row_add, const AderDrop* drop, void* stream);
int ader_gemm_atb_x3(const float* A, const float* G, float* slab, float* dW, float* db, int M, int H, void* stream);
/* The same products batched: n <= 16 independent (A[i], G[i], M[i]) -> (dW[i], db[i] or NULL) in one product launch + one
 * reduce launch (all weight gradients of a backward pass, tf.gradients of modules.py:172-174,254-261).  A/G/dW/db/M are HOST
 * arrays (of device pointers).  slab: ader_gemm_atb_batch_slabs(M, n) * 160 * 160 floats of scratch. */
int ader_gemm_atb_batch_slabs(const int* M, int n);
int ader_gemm_atb_x3_batch(const float* const* A, const float* const* G, float* const* dW, float* const* db, const int* M, int n,
                           float* slab, int H, void* stream);
/* ---- whole forward stack of a session in one workgroup (seq_fwd.hip) ------------------------------------------------
 * Replaces, for heads == 1, T <= 64, even H <= 150 and <= ADER_SEQ_MAXL blocks, the chain embed_fwd -> per block [ln_fwd,
 * gemm_x3 x3, attn_x3_fwd, ln_fwd, gemm_x3 x2] -> ln_fwd of the reference forward (ADER.py:41-91; modules.py:44-48,
 * 118-130,172-223,254-266) by a single launch of B workgroups.  Writes exactly the activations those kernels write (same
 * buffers and layouts; a block with pruned != 0 keeps only position T-1 of its query / FFN path in compact [B,H] /
 * [B] / [B,T] buffers, K, V and kmask stay [B*T,..]).  w[i]: the 4 prepared planes (ader_wprep) of wq, wk, wv, w1, w2. */
#define ADER_SEQ_MAXL 4

typedef struct {
    const void* w[5];
    const float* bias[5];
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    float *q_in, *mean1, *std1, *kmask, *qmask, *Q, *K, *V, *P, *x1, *y, *mean2, *std2, *h1d, *x2;
    AderDrop d_attn, d_ffn1, d_ffn2;
    int pruned, pad_;
} AderSeqBlock;
typedef struct {
    const int* seq;               /* [B,T] item ids, 0 = padding */
    const float *emb, *pos;       /* item table [V,H], positional table [T,H] */
    float* x0;                    /* [B*T,H] block-0 input (saved for backward) */
    int* status;                  /* device status word (bad item id) */
    const float *lnf_g, *lnf_b;
    float *rep, *meanf, *stdf;    /* [B,H], [B], [B] */
    int B, T, H, V, L;
    float sqrtH, sqrt_dh;
    int pad_;
    AderDrop d_emb;
    AderSeqBlock blk[ADER_SEQ_MAXL];
} AderSeqFwd;
int ader_seq_fwd(const AderSeqFwd* desc, void* stream);

/* ---- the same stack on PACKED session tiles (seqp_plan.hip, seqp_fwd.hip, seqp_bwd.hip) ---------------------------------
 * The reference pads every session to maxlen (util.py:161-169) and computes the padding too; a padded position influences no real
 * one (key masked modules.py:188-193, output re-zeroed ADER.py:80, zero gradient), so these kernels drop it: ader_seq_pack_plan
 * lays the REAL positions of a batch out in 64-row tiles (several short sessions per tile, a session never split, batch order kept
 * inside a tile) and ader_seqp_fwd / ader_seqp_bwd_* run the unchanged arithmetic of ader_seq_fwd / ader_seq_bwd_* per tile,
 * attention block-diagonal over the sessions of a tile.  Activation tensors of these kernels are in TILE order ([tile*64 + row, ..],
 * probabilities [tile][key][query]); rep / meanf / stdf [B,..], the compact tensors of a pruned block [B,..] and the gradient rows of
 * the input embeddings [B*T,H] (only real positions written) stay session-indexed.
 * Plan arrays (device, written by ader_seq_pack_plan): hdr [8] = {tiles, 64*tiles, real positions, window, -, -, -, ticket of the
 * length pass: zero before the first call}; tile_rows [<= B]; per
 * packed row (<= 64 B entries): ids (item id), lpos (b*T + t), gpos (GLOBAL position (b + row offset)*T + t: the dropout counters
 * of a data-parallel rank, see AderDrop), info = first tile row of the row's session | last-position flag << 6 | t << 8 | b << 16;
 * per session: srow0 (first packed row), slen (real positions, >= 1: an all-padding session keeps position T-1).
 * row0 / split_rows / row0_ex: global row of local session 0, first local exemplar session (-1: none) and its global row.
 * w1_min / w1_max / target: stream window of the short (<= 16 positions) class, at most 49 -- shrunk towards w1_min so that the
 * batch fills about `target` tiles (0: always w1_max). */
typedef struct {
    int *hdr, *tile_rows, *ids, *lpos;
    unsigned* gpos;
    int *info, *srow0, *slen;
} AderSeqPack;
int ader_seq_pack_plan(const int* seq, int B, int T, int row0, int split_rows, int row0_ex, int w1_min, int w1_max, int target,
                       const AderSeqPack* out, void* stream);
/* max_tiles: host-side upper bound of the tile count = grid size (<= 0 or > B: B); workgroups beyond hdr[0] exit at once */
int ader_seqp_fwd(const AderSeqFwd* desc, const AderSeqPack* pack, int max_tiles, void* stream);

/* ---- session-tiled backward chains of one block (seq_bwd.hip) --------------------------------------------------------
 * The row-local kernels on either side of the attention backward, one launch of B workgroups each (tf.gradients of
 * ADER.py:62-81).  Tensors and layouts as written by the per-op kernels they replace; pruned != 0: the block kept only
 * position T-1 of its query / FFN path (compact [B,H] / [B] tensors: dx2, h1d, x1, mean2, std2, dh2, da, dx1, dQ, mean1,
 * std1); K/V-side tensors (dK, dV, x, dx) are always [B*T,H].  slab: [B][2][H] per-session partial sums of the LayerNorm
 * gamma / beta gradients (reduce with ader_reduce_slabs).  w*: prepared planes (ader_wprep).  */
typedef struct {      /* replaces ader_mask_dropgrad, ader_gemm_x3<RELUDROPGRAD,trans>, ader_gemm_x3<ADD,trans>, ader_ln_bwd */
    const int* seq;                       /* [B,T] */
    const float *dx2, *h1d, *x1, *mean2, *std2, *ln2_g;
    const void *w2, *w1;
    float *dh2, *da, *dx1, *slab;
    AderDrop d_ffn1, d_ffn2;
    int B, T, H, pruned;
} AderSeqBwdFfn;
typedef struct {      /* replaces ader_gemm_x3<ADD,trans> x3, ader_ln_bwd, ader_add_rows (and ader_embed_bwd_rows if emb_bwd) */
    const int* seq;                       /* [B,T]; read only when emb_bwd != 0 */
    const float *dQ, *dx1, *dK, *dV, *x, *mean1, *std1, *ln1_g;
    const void *wq, *wk, *wv;
    float *dx, *slab;                     /* dx: [B*T,H] gradient of the block input */
    AderDrop d_emb;                       /* emb_bwd: also apply the prologue backward dx *= (seq != 0) * keep * scale */
    int B, T, H, pruned, emb_bwd, pad_;
} AderSeqBwdQkv;
int ader_seq_bwd_ffn(const AderSeqBwdFfn* desc, void* stream);
int ader_seq_bwd_qkv(const AderSeqBwdQkv* desc, void* stream);
/* ---- packed session tiles, backward (seqp_bwd.hip) */
/* backward chains of one block on the tiles (descriptors of ader_seq_bwd_ffn / _qkv below; `seq` unused -- the ids come from the
 * plan; slab: [max_tiles][2][H] per-TILE partial sums; emb_bwd: dx is the session-indexed [B*T,H] tensor, real positions only) */
int ader_seqp_bwd_ffn(const AderSeqBwdFfn* desc, const AderSeqPack* pack, int max_tiles, void* stream);
int ader_seqp_bwd_qkv(const AderSeqBwdQkv* desc, const AderSeqPack* pack, int max_tiles, void* stream);
/* attention backward on a tile (ader_attn_x3_bwd; heads == 1; PT [tile][key][query] as ader_seqp_fwd stored it) and of a pruned
 * block (ader_attn_last_bwd: Q_last / dO_last / dQ_last / P_last / qmask_last compact [B,..], K / V / dK / dV / kmask tile order) */
int ader_attnp_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* PT, const float* kmask,
                   const float* qmask, float* dQ, float* dK, float* dV, int B, int T, int H, const AderDrop* drop,
                   const AderSeqPack* pack, int max_tiles, void* stream);
int ader_attnp_last_bwd(const float* dO_last, const float* Q_last, const float* K, const float* V, const float* P_last,
                        const float* kmask, const float* qmask_last, float* dQ_last, float* dK, float* dV, int B, int T, int H,
                        const AderDrop* drop, const AderSeqPack* pack, void* stream);
/* dpos [T,H] = sum over the sessions that have position t of dx[b*T + t] (gradient of the positional table, modules.py:118-130
 * differentiated) from session-indexed rows whose padding rows were never written */
int ader_pos_grad_packed(const float* dx, const int* slen, float* dpos, int B, int T, int H, void* stream);
/* ader_gemm_atb_x3_batch with operands in tile order: Mplan (host, or NULL = M) shares the workgroups out by the rows expected to
 * exist, Mdev[i] (device: 64 x tiles) bounds M[i], trows[i] (device: rows per tile) masks the unwritten rows; NULL entries: plain */
int ader_gemm_atb_x3_batch_pk(const float* const* A, const float* const* G, float* const* dW, float* const* db, const int* M,
                              const int* Mplan, const int* const* Mdev, const int* const* trows, int n, float* slab, int H, void* stream);
/* g = dx2*(seq!=0); dh2 = g*keep*scale : backward entry of modules.py:262-266 + ADER.py:80 */
int ader_mask_dropgrad(const float* dx2, const int* seq, float* g, float* dh2, int rows, int H, int row_mul, int row_add,
                       const AderDrop* drop, void* stream);
/* dst[(r*row_mul+row_add), :] += src[r, :] */
int ader_add_rows(const float* src, float* dst, int rows, int H, int row_mul, int row_add, void* stream);

/* ---- attention core: modules.py:177-223 --------------------------------------------------------------- */
int ader_attn_fwd(const float* Q, const float* K, const float* V, const float* q_in, const float* kmask,
                  const float* qmask, float* out, float* P, int B, int T, int H, int heads, const AderDrop* drop, void* stream);
int ader_attn_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* P, const float* kmask,
                  const float* qmask, float* dQ, float* dK, float* dV, int B, int T, int H, int heads,
                  const AderDrop* drop, void* stream);

/* bf16x3 variants (hi/lo operand split on v_mfma_f32_32x32x16_bf16, float32-grade accuracy); dh = H/heads even.
 * PT is stored transposed ([B,heads,key,query]) and is private to this fwd/bwd pair. */
int ader_attn_x3_fwd(const float* Q, const float* K, const float* V, const float* q_in, const float* kmask,
                     const float* qmask, float* out, float* PT, int B, int T, int H, int heads, const AderDrop* drop, void* stream);
int ader_attn_x3_bwd(const float* dO, const float* Q, const float* K, const float* V, const float* PT, const float* kmask,
                     const float* qmask, float* dQ, float* dK, float* dV, int B, int T, int H, int heads,
                     const AderDrop* drop, void* stream);

/* Last block: only position T-1 feeds the representation (ADER.py:85), so its attention needs one query row per sequence
 * (exact).  Q_last, q_in_last, out_last, dQ_last: [B,H]; K, V, dK, dV: [B,T,H]; P_last: [B,heads,T]; qmask_last: [B]. */
int ader_attn_last_fwd(const float* Q_last, const float* K, const float* V, const float* q_in_last, const float* kmask,
                       const float* qmask_last, float* out_last, float* P_last, int B, int T, int H, int heads,
                       const AderDrop* drop, void* stream);
int ader_attn_last_bwd(const float* dO_last, const float* Q_last, const float* K, const float* V, const float* P_last,
                       const float* kmask, const float* qmask_last, float* dQ_last, float* dK, float* dV, int B, int T,
                       int H, int heads, const AderDrop* drop,
                       void* stream);

/* ---- full-catalog logits + loss: ADER.py:88-93, 108-137 ------------------------------------------------ */
/* Row descriptors, all [Bp]: lab (1-based target item or 0), ncol (valid columns: N, or Np for distilled rows, 0 for
 * padding rows), wrow (loss weight), trow (row of `teacher` or -1), tlse (log-sum-exp of that teacher row). */
int ader_logits_sub(int N);
int ader_logits_parts(int N);
int ader_logits_ranges(int N, int Bp);
/* Builds the row descriptors of one step on the device (train rows first, exemplar rows after, ADER.py:113-118). */
int ader_build_rowinfo(const int* pos, int n_train, const int* ex_pos, const int* ex_trow, int n_ex, int N, int Np,
                       float w_train, float w_ex, int Bp, int* lab, int* ncol, float* wrow, int* trow, void* stream);
int ader_row_lse(const float* x, long ld, int ncols, const int* rows, int nrows, float* out, void* stream);
int ader_logits_loss_fwd(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* lab, const int* ncol,
                         const float* wrow, const int* trow, const float* tlse, const float* teacher, long ldt,
                         float* part, float* lse, float* rowloss, float* loss, void* stream);
/* backward part 1: drep [B,H] = dlogit . E   (slab: ader_logits_ranges(N,Bp)*Bp*160 floats of scratch) */
int ader_logits_bwd_drep(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* lab, const int* ncol,
                         const float* wrow, const int* trow, const float* tlse, const float* teacher, long ldt,
                         const float* lse, float* slab, float* drep, void* stream);
/* backward part 2: demb rows 1..N = dlogit^T . rep  (overwritten, each row written once) */
int ader_logits_bwd_demb(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* lab, const int* ncol,
                         const float* wrow, const int* trow, const float* tlse, const float* teacher, long ldt,
                         const float* lse, float* demb, void* stream);
/* out[b, 0:N] = rep[b] . E[1..N]^T  (model.logits fetch util.py:452; teacher logits util.py:433) */
int ader_logits_store(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* ncol_all, float* out,
                      long ldo, void* stream);
/* 0-based rank of target[b] among items 1..N, ties -> lower index first: replaces argsort(argsort(-logits)) (ADER.py:103)
 * + the host gather pred[label-1] (util.py:325). */
int ader_rank_targets(const float* rep, const float* emb, int B, int Bp, int H, int N, const int* target, const int* ncol,
                      float* tlogit, int* rank, void* stream);

/* ---- bf16-MFMA variant of the one-hot softmax CE (fp32 master table, fp32 accumulate/softmax): ADER.py:88-93 ------ */
/* Bp % 128 == 0, H even.  Scratch: rep_bf Bp*168 bf16; pm, pl: R*Bp floats; pO: R*Bp*160 floats, R = ader_lbf_ranges(N,Bp).
 * Outputs: lse/off/rowloss [Bp], loss [1], drep [B,H] (complete: includes the one-hot target term). */
/* `shadow`: bf16 copy of the table, [item_num+1][168] (row stride 336 B, columns >= H zero), kept in sync by
 * ader_adam_step and (re)built by ader_lbf_shadow_refresh after initialisation / checkpoint load. */
int ader_lbf_shadow_refresh(const float* emb, void* shadow, size_t rows, int H, void* stream);
int ader_lbf_ranges(int N, int Bp);
int ader_lbf_fwd(const float* rep, const void* shadow, int item_num, int B, int Bp, int H, int N, const int* lab,
                 const float* wrow, void* rep_bf, float* pm, float* pl, float* pO, float* lse, float* off, float* rowloss,
                 float* loss, float* drep, void* stream);
/* Forward of a DISTILLED step (ADER.py:108-137: loss = CE_train + lambda * mean_e(-sum_j softmax(teacher_e)_j log_softmax(
 * logits_e[:Np])_j)) on the bf16 flash path.  rep: compact [n_train + n_ex, H], exemplar rows last (main.py:229).  Inside, rows are
 * laid out [train rows padded to 128 | exemplar rows padded to 128] (Bp rows, kd_row0 = first exemplar row): lab / wrow / trow / tlse2
 * (written here) and lse / off / rowloss are [Bp] in that layout; drep is compact.  ex_trow[e]: row of `teacher` [*, ldt] for
 * exemplar e; tlse_all[r]: natural log-sum-exp of teacher row r over [0, Np).  w_train = 1/B_train, w_ex = lambda/B_ex.
 * Scratch: rep_bf Bp*168 bf16; R = ader_lbf_ranges_kd(N, Bp, kd_row0); pm, pl: R*Bp; pO: R*Bp*160; pO2: R2*(Bp-kd_row0)*160 floats with
 * R2 = ader_lbf_readout_ranges(N, Bp, kd_row0) (the teacher-readout chunks have their own, finer partition of the item blocks). */
int ader_lbf_fwd_kd(const float* rep, const void* shadow, int item_num, int n_train, int n_ex, int kd_row0, int Bp, int H, int N,
                    int Np, const int* pos, const int* ex_trow, const float* teacher, long ldt, const float* tlse_all,
                    float w_train, float w_ex, int* lab, float* wrow, int* trow, float* tlse2, void* rep_bf, float* pm, float* pl,
                    float* pO, float* pO2, float* lse, float* off, float* rowloss, float* loss, float* drep, void* stream);
int ader_lbf_ranges_kd(int N, int Bp, int kd_row0);
int ader_lbf_readout_ranges(int N, int Bp, int kd_row0);
/* the same forward at float32 grade (x3): the fp32 table instead of the shadow, two operand planes rep_hi / rep_lo.  Scratch:
 * pm / pl / pO with R = ader_lbf_ranges(N, Bp); pO2: ader_lx3_readout_ranges(Np, Bp - kd_row0) * (Bp - kd_row0) * 160 floats (the
 * teacher readout is a launch of its own here) */
int ader_lx3_readout_ranges(int Np, int Bk);
int ader_lx3_fwd_kd(const float* rep, const float* emb, int item_num, int n_train, int n_ex, int kd_row0, int Bp, int H, int N,
                    int Np, const int* pos, const int* ex_trow, const float* teacher, long ldt, const float* tlse_all,
                    float w_train, float w_ex, int* lab, float* wrow, int* trow, float* tlse2, void* rep_hi, void* rep_lo,
                    float* pm, float* pl, float* pO, float* pO2, float* lse, float* off, float* rowloss, float* loss, float* drep,
                    void* stream);
/* Pieces of ader_lbf_fwd for catalog-sharded data parallelism (each rank holds 1/W of the table rows and streams only
 * those; ADER.py:91-93 with the item axis split across ranks): ader_lbf_prep builds the bf16 operand rows [Bp,168] of the
 * (all-gathered) representations; ader_lbf_fwd_shard returns per batch row the softmax partials {max (log2 domain), sum,
 * weighted table-row sum[H]} over the items [item_begin+1, item_begin+item_count] (clipped to N) in part [Bp][152]; the
 * ranks' partials are merged on the host side of the ABI (ader_amd/engine.py). */
int ader_lbf_prep(const float* rep, void* rep_bf, int B, int Bp, int H, void* stream);
/* merge of the ranks' partials of THIS rank's rows (parts [world][Bp][152], slice i from rank i) -> lse/off/rowloss [Bp],
 * loss [1], drep [B,H]; e_lab [B,H] = fp32 table rows of the labels (ADER.py:88-93 evaluated over a sharded item axis) */
int ader_lbf_merge_parts(const float* parts, int world, int Bp, int B, int H, const float* e_lab, const void* rep_bf,
                         const float* wrow, float* lse, float* off, float* rowloss, float* loss, float* drep, void* stream);
/* table rows travelling between ranks: out[p] = table[ids[p]] if lo < ids[p] <= hi else 0;  and the receiving side:
 * recv [world][n][H] (slice i from rank i) -> table[ids[p]] for p < n_tab, extra[p - n_tab] for the rest (label rows);
 * the owner of id is (id-1)/shard; padding ids (0) are skipped (tf.nn.embedding_lookup of modules.py:127 across shards) */
int ader_gather_owned(const float* table, const int* ids, int n, int H, int lo, int hi, float* out, void* stream);
int ader_scatter_owned(const float* recv, const int* ids, int n, int n_tab, int H, int shard, int world, float* table,
                       float* extra, void* stream);
int ader_lbf_fwd_shard(const void* rep_bf, const void* shadow, int item_num, int Bp, int H, int N, int item_begin,
                       int item_count, float* pm, float* pl, float* pO, float* part, void* stream);
/* ... and at float32 grade (x3): the same two steps streaming the fp32 table rows of the shard (operand planes rep_hi / rep_lo of
 * the all-gathered representations from ader_lx3_prep; the merge takes the fp32 representations and label rows). */
int ader_lx3_fwd_shard(const void* rep_hi, const void* rep_lo, const float* emb, int item_num, int Bp, int H, int N,
                       int item_begin, int item_count, float* pm, float* pl, float* pO, float* part, void* stream);
/* ... distilled rows (ADER.py:132-137) under the same scheme: the teacher readout O2 = sum_j softmax(teacher)_j E_j over the rank's
 * item shard for the Bk gathered exemplar rows (part2 [Bk][152], channels at [2, 2 + H); trow [Bk] teacher row or -1, tlse2 [Bk] the
 * teacher's log2-domain log-sum-exp over [0, Np); scratch pO2: ader_lx3_readout_ranges(item_count, Bk) * Bk * 160 floats), and the
 * merge of the W ranks' student partials (ader_lx3_fwd_shard called with N = Np) and readout sums for THIS rank's exemplar rows:
 * loss row w (lse - rep . O2), dRep = w (O1 / l - O2), lse, backward offset. */
int ader_lx3_readout_shard(const float* emb, int item_num, int Bk, int H, int Np, int item_begin, int item_count, const float* teacher,
                           long ldt, const int* trow, const float* tlse2, float* pO2, float* part2, void* stream);
int ader_lx3_merge_parts_kd(const float* parts_s, const float* parts_t, int world, int Bk, int B, int H, const float* rep,
                            const float* wrow, float* lse, float* off, float* rowloss, float* drep, void* stream);
int ader_lx3_merge_parts(const float* parts, int world, int Bp, int B, int H, const float* e_lab, const float* rep,
                         const float* wrow, float* lse, float* off, float* rowloss, float* loss, float* drep, void* stream);
/* float32-grade variant of ader_lbf_fwd ("x3": every product as three bf16 MFMAs on hi/lo operand splits, fp32 accumulate;
 * ADER.py:91-93 is fp32 arithmetic).  Streams the fp32 table itself -- no bf16 shadow exists in this mode.  rep_hi/rep_lo:
 * Bp*168 bf16 each (written here: bf16(rep), bf16(rep - hi)); the other scratch and the outputs as in ader_lbf_fwd. */
int ader_lx3_prep(const float* rep, void* rep_hi, void* rep_lo, int B, int Bp, int H, void* stream);
int ader_lx3_fwd(const float* rep, const float* emb, int item_num, int B, int Bp, int H, int N, const int* lab,
                 const float* wrow, void* rep_hi, void* rep_lo, float* pm, float* pl, float* pO, float* lse, float* off,
                 float* rowloss, float* loss, float* drep, void* stream);
/* ... with the backward of the final LayerNorm fused into the merge launch: rep = LN_f(x) (ADER.py:83-85), so the workgroup that forms a
 * row of dRep also forms dx = LN_f'(dRep) for it (bit-equal to ader_ln_bwd) and the row's gamma / beta partials.  x [B,H] the input of
 * the final LayerNorm (compact rows, as rep), mean / std [B] its statistics, gamma [H]; dx [B,H] out; slab [B][2][H] out (reduce with
 * ader_reduce_slabs(slab, 2 H, B, H, 1, H, dgamma, dbeta)).  lnf == NULL: exactly ader_lx3_fwd_img / ader_lx3_fwd_kd (loss may be NULL
 * in both: ader_lbf_sum(rowloss, ...) later).  rep_img of the distilled form: as in ader_lx3_fwd_img -- the operand images of
 * ader_tab_update_x3_kd over the padded row layout, cut by the launch that cuts the planes (NULL: ader_x3_rep_image later). */
typedef struct { const float *x, *mean, *std, *gamma; float *dx, *slab; } AderLnfBwd;
int ader_lx3_fwd_img_lnf(const float* rep, const float* emb, int item_num, int B, int Bp, int H, int N, const int* lab,
                         const float* wrow, void* rep_hi, void* rep_lo, float* pm, float* pl, float* pO, float* lse, float* off,
                         float* rowloss, float* loss, float* drep, void* rep_img, const AderLnfBwd* lnf, void* stream);
int ader_lx3_fwd_kd_lnf(const float* rep, const float* emb, int item_num, int n_train, int n_ex, int kd_row0, int Bp, int H, int N,
                        int Np, const int* pos, const int* ex_trow, const float* teacher, long ldt, const float* tlse_all,
                        float w_train, float w_ex, int* lab, float* wrow, int* trow, float* tlse2, void* rep_hi, void* rep_lo,
                        float* pm, float* pl, float* pO, float* pO2, float* lse, float* off, float* rowloss, float* loss, float* drep,
                        void* rep_img, const AderLnfBwd* lnf, void* stream);
/* loss = sum of rowloss[0..n) in a fixed order: the last launch of ader_lx3_fwd when that was given loss = NULL (reference
 * ADER.py:93: reduce_mean of the per-row cross entropies; the weights 1/B are already in rowloss). */
int ader_lbf_sum(const float* rowloss, int n, float* loss, void* stream);
/* ader_lx3_fwd that also writes the operand images of the fused update (rep_img: ader_x3_rep_image_bytes(Bp) bytes, zero-initialised
 * once, 16-byte aligned; NULL: exactly ader_lx3_fwd) -- ader_x3_rep_image need not be launched for this batch. */
int ader_lx3_fwd_img(const float* rep, const float* emb, int item_num, int B, int Bp, int H, int N, const int* lab,
                     const float* wrow, void* rep_hi, void* rep_lo, float* pm, float* pl, float* pO, float* lse, float* off,
                     float* rowloss, float* loss, float* drep, void* rep_img, void* stream);
/* Gradient of the one-hot softmax CE w.r.t. the item table (ADER.py:91-93 differentiated): demb rows 1..N overwritten
 * (each row written once, then the sparse one-hot term is added with float atomics).  The GEMM operand is cut from the
 * fp32 table `emb` inside the kernel (bf16, or hi/lo when rep_lo != NULL: x3 mode). */
int ader_tab_grad(const void* rep_hi, const void* rep_lo, const float* emb, int item_num, int B, int Bp, int H, int N,
                  const int* lab, const float* wrow, const float* off, float* demb, void* stream);

/* ... for a distilled step (padded layout of ader_lbf_fwd_kd / ader_lx3_fwd_kd; rows [kd_row0, Bp) carry the teacher term of
 * ADER.py:132-137): the gradient-only form behind the dense data-parallel exchange. */
int ader_tab_grad_kd(const void* rep_hi, const void* rep_lo, const float* emb, int item_num, int Bp, int kd_row0, int H, int N,
                     int Np, const int* lab, const float* wrow, const float* off, const float* teacher, long ldt, const int* trow,
                     const float* tlse2, float* demb, void* stream);

/* Fused table update at float32 grade (x3), csrc/table_update_x3.hip: table-gradient GEMM + sparse terms (input-embedding rows sp_*,
 * one-hot targets tg_*, both sorted by item id) + tf.train.AdamOptimizer (ADER.py:96) on table rows 1..N of emb / adam_m / adam_v in ONE
 * pass; the table gradient is never written to memory.  lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t).  extra_grad: dense fp32 gradient
 * [item_num+1, H] added row by row before the update, or NULL.  Bucket layout of the sorted lists: bucket j covers ids
 * [g*j + id0, g*(j+1) + id0) with g = ader_fused_bucket_gran(), id0 = ader_fused_bucket_id0().  rep_img: the operand rows rearranged into
 * the bank-conflict-free LDS images that the kernel streams by LDS-DMA (ader_x3_rep_image from the two planes rep_hi / rep_lo
 * [Bp,168]; ader_x3_rep_image_bytes(Bp) bytes, 16-byte aligned; Bp % 32 == 0).  Replaces the dense-Adam + table-gradient op
 * sites ADER.py:91-96 for the item table. */
int ader_x3_rep_image_bytes(int Bp);
/* kernel choice between the update's two kernels: catalogs of more than `tiles` 64-row tiles take a pair of tiles per workgroup (k_tab32x3),
 * smaller ones a single tile (k_tab16x3); default 0 = pairs always; negative: query only.  Returns the previous value. */
int ader_x3_update_pair_min_tiles(int tiles);
int ader_x3_rep_image(const void* rep_hi, const void* rep_lo, int Bp, void* img, void* stream);
int ader_tab_update_x3(const void* rep_hi, const void* rep_lo, const void* rep_img, int item_num, int B, int Bp, int H, int N,
                       const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src, float sp_scale,
                       const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta, const float* wrow, float* emb,
                       float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                       int tile_count, const float* extra_grad, void* stream);
int ader_tab_update_x3_kd(const void* rep_hi, const void* rep_lo, const void* rep_img, int item_num, int Bp, int kd_row0, int H,
                          int N, int Np, const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src,
                          float sp_scale, const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta,
                          const float* wrow, const float* teacher, long ldt, const int* trow, const float* tlse2, float* emb,
                          float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, void* stream);
/* ... restricted to the 128-item tiles [tile_begin, tile_begin + tile_count) (a rank's shard of a catalog-sharded table; Bp, kd_row0
 * and the per-row arrays describe the GLOBAL batch [all train rows | all exemplar rows]; tile_count < 0: all tiles) */
int ader_tab_update_x3_kd_range(const void* rep_hi, const void* rep_lo, const void* rep_img, int item_num, int Bp, int kd_row0, int H,
                          int N, int Np, const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src,
                          float sp_scale, const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta,
                          const float* wrow, const float* teacher, long ldt, const int* trow, const float* tlse2, float* emb,
                          float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                                int tile_count, void* stream);
/* The bf16-mode form over 128-row tiles: the GEMM operand is the tile's bf16 shadow rows (`shadow` is read AND rewritten) and the
 * sorted lists are addressed through their 64-id bucket offsets sp_start / tg_start.  Faster than ader_tab_update(rep_lo = NULL) at
 * H = 150 on MI355X although it reads 336 B more per row (measurements: DESIGN.md). */
int ader_tab_update_sh(const void* rep_bf, void* shadow, int item_num, int B, int Bp, int H, int N, const float* off,
                       const int* sp_ids, const int* sp_rows, const int* sp_start, int n_sp, const float* sp_src,
                       float sp_scale, const int* tg_ids, const int* tg_rows, const int* tg_start, int n_tg,
                       const float* wrow, float* emb, float* adam_m, float* adam_v, float lr_t, float beta1, float beta2,
                       float eps, int tile_begin, int tile_count, const float* extra_grad, void* stream);
/* ... and for a DISTILLED step (ADER.py:132-137): batch rows [kd_row0, Bp) of the padded layout of ader_lbf_fwd_kd are exemplar
 * rows whose dlogit is w (softmax(s[:Np]) - softmax(teacher row)); wrow / off / trow / tlse2 as that call left them. */
int ader_tab_update_sh_kd(const void* rep_bf, void* shadow, int item_num, int Bp, int kd_row0, int H, int N, int Np,
                          const float* off, const int* sp_ids, const int* sp_rows, const int* sp_start, int n_sp,
                          const float* sp_src, float sp_scale, const int* tg_ids, const int* tg_rows, const int* tg_start,
                          int n_tg, const float* wrow, const float* teacher, long ldt, const int* trow, const float* tlse2,
                          float* emb, float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps,
                          void* stream);
/* tile_meta: per 64-row tile the list record {k0, k1, first 8 (id, row) entries} x 2 lists, ader_tab_meta_ints(N) ints, built
 * by ader_tab_tile_meta from the bucket offsets sp_start / tg_start (one coalesced read per tile inside the update). */
int ader_tab_meta_ints(int N);
int ader_tab_tile_meta(const int* sp_ids, const int* sp_rows, const int* sp_start, const int* tg_ids, const int* tg_rows,
                       const int* tg_start, int N, int* rec, void* stream);
/* tile_begin/tile_count: restrict the update to 128-item tiles [tile_begin, tile_begin+tile_count) (row-sharded table
 * update under data parallelism; tile_count < 0 = all tiles).  B/Bp then describe the GLOBAL batch. */
/* The lists themselves (index work, bit-exact): the input positions seq [n_sp] and the labels lab [n_tg] grouped by bucket
 * [g j + id0, g (j+1) + id0) in id order, inside a bucket by (id, POSITION) (every table row receives its contributions as one
 * run in position order: bit-reproducible; padding entries, id 0, are left out), their ids and positions, and the bucket offsets
 * start[j], j = 0 .. ceil(N/g) (ader_sparse_lists_starts(N) ints per list).  A counting sort (integer atomics) whose arrival
 * order is replaced by position ranks.  scratch: ader_sparse_lists_scratch_n(n_sp, n_tg, N) ints.  Replaces, for the gradient
 * of tf.nn.embedding_lookup (modules.py:127) and of the one-hot labels (ADER.py:89), TF's unsorted_segment_sum by a
 * deterministic row-ordered accumulation inside the table update. */
int ader_sparse_lists_scratch_n(int n_sp, int n_tg, int N);
int ader_sparse_lists_starts(int N);
int ader_sparse_lists(const int* seq, int n_sp, const int* lab, int n_tg, int N, int* scratch, int* sp_ids, int* sp_rows,
                      int* sp_start, int* tg_ids, int* tg_rows, int* tg_start, void* stream);
/* ... followed by ader_tab_tile_meta (rec: ader_tab_meta_ints(N) ints; NULL: the lists alone) -- ONE launch when the problem fits one
 * workgroup's LDS (at most 131,072 items and 131,072 entries: the shipped datasets), the same chain of launches otherwise; same
 * outputs either way. */
int ader_sparse_lists_meta(const int* seq, int n_sp, const int* lab, int n_tg, int N, int* scratch, int* sp_ids, int* sp_rows,
                           int* sp_start, int* tg_ids, int* tg_rows, int* tg_start, int* rec, void* stream);
int ader_fused_bucket_gran(void);
int ader_fused_bucket_id0(void);

/* ---- EWC baseline (reference EWC.py:115-164) -------------------------------------------------------------------- */
/* Fisher accumulation F += scale * g * g over a flat buffer (EWC.py:160-163), and the penalty of EWC.py:121-124:
 * loss[0] += lambda/2 * sum F (theta - prev)^2, grad += lambda * F * (theta - prev).  part: 1024 floats of scratch. */
int ader_sq_accum(const float* g, float* F, size_t n, float scale, void* stream);
int ader_ewc_penalty(const float* theta, const float* prev, const float* F, float* grad, size_t n, float lambda_, float* part,
                     float* loss, void* stream);

/* ---- optimiser: tf.train.AdamOptimizer (ADER.py:96), dense over one flat buffer ------------------------ */
/* shadow (optional, may be NULL): bf16 shadow of the first table_elems parameters (the item table, rows of H), see above */
int ader_adam_step(float* p, float* m, float* v, const float* g, size_t n, float lr_t, float beta1, float beta2, float eps,
                   void* shadow, size_t table_elems, int H, void* stream);
int ader_fill(float* p, size_t n, float value, void* stream);
int ader_reduce_slabs(const float* src, long slab_stride, int S, int ld, int n_rows, int n_cols, float* dst,
                      float* dst_extra, void* stream);
/* up to 8 such reductions in one launch (arrays of n job descriptions; same arithmetic and summation order per job) */
int ader_reduce_slabs_batch(const float* const* src, const long* slab_stride, const int* S, const int* ld, const int* n_rows,
                            const int* n_cols, float* const* dst, float* const* dst_extra, int n, void* stream);

/* ---- data-parallel catalog-sharded step: bookkeeping of the packed row exchange in one launch (csrc/pack_plan.hip).  Serves the
 *      gather of modules.py:127 and its gradient when the item table is sharded over the ranks (the reference is single-device,
 *      main.py:96).  ids_g [W][n_all] gathered ids (n_pos input positions, then labels, per rank); rank r owns items
 *      [1 + r shard_items, (r + 1) shard_items].  cnt [2][W][W] ints (owner x destination counts: all positions / input positions);
 *      send_id / ids_back (<= W n_all), perm (n_all), back_src (<= n_pos): int64 index arrays, see the file header.  W <= 16. */
int ader_pack_plan(const int* ids_g, int W, int n_all, int n_pos, int rank, int shard_items, int* cnt, long* send_id, long* ids_back,
                   long* perm, long* back_src, void* stream);

/* ---- herding exemplar selection: util.py:401-434 (the loop of ExemplarGenerator.herding, called per label from
 *      herding_selection util.py:447-457) -- ALL label groups of a period in one call ------------------------------------
 * rep [n_total,H] candidate representations in group order; seg [G+1] group offsets; quota [G]; max_steps [G] = number of
 * integers k < 1.1*min(quota,n) (the reference's loop bound, evaluated by the host in float64).  Scratch: D n_total*H + G + 64
 * floats, chosen n_total bytes.  Out: sel [n_total] (per group, the selected LOCAL indices in selection order at the start of
 * its span), sel_cnt [G], steps_out [G] (optional).  H = 150 (the reference's hidden_units): register-resident kernel;
 * any other H <= 256: the generic kernel.  Both follow the canonical float32 spec of oracle/herding_ref.py bit for bit. */
int ader_herding_select(const float* rep, const long* seg, const int* quota, const int* max_steps, int G, long n_total,
                        int H, float* D, unsigned char* chosen, int* sel, int* sel_cnt, int* steps_out, void* stream);
/* ---- device-side feeder: util.py:218-262 (Sampler.sampler / exemplar_sampler: the rows of a batch by the shuffled index list) and
 *      main.py:229 (exemplar rows appended to the train rows), as ONE launch over the GPU-resident packed rows ----------------------
 * rows_t / rows_e [*, T+1]: packed rows of the train / exemplar Sampler (inputs right-aligned in zeros, label last); idx_t [n_t] /
 * idx_e [n_e]: int64 row indices of this batch.  Writes seq [Bt + Be, T] = [n_t train rows | zero rows up to Bt | n_e exemplar rows |
 * zero rows up to Be], pos [Bt] (label, 0 = padding row: weight 0 in every loss kernel), ex_pos [Be] (exemplar labels, optional) and
 * ex_trow [Be] (idx_e as int32 = the exemplar's teacher row, -1 = padding row; optional). */
int ader_feed_step(const int* rows_t, const long* idx_t, int n_t, int Bt, const int* rows_e, const long* idx_e, int n_e, int Be, int T,
                   int* seq, int* pos, int* ex_pos, int* ex_trow, void* stream);
/* out [na + nb] = a | b (the label list of a one-hot replay step: train labels, then exemplar labels, ADER.py:126-131) */
int ader_concat_i32(const int* a, int na, const int* b, int nb, int* out, void* stream);

/* ---- native step driver: main.py:220-256 -- the reference runs ONE sess.run(train_op) per step; the executor, not Python, walks the
 *      ops.  A plan is the launch sequence of one (shape, mode) of the step: launcher calls on two lanes (0 = main stream, 1 = side
 *      stream) and the event edges between the lanes; ader_step_enqueue walks it in ONE call (csrc/step_plan.hip) --------------------
 * Slots: every launcher argument as one uint64 -- pointers as addresses, int / unsigned / long / size_t sign-extended, float as its
 * IEEE bits in the low 32.  The launcher's trailing `stream` argument is a slot too and is filled with the lane's stream per step.
 * Host descriptors (arguments that are HOST pointers: AderSeqFwd, AderDrop, pointer arrays ...) are listed as blobs and copied into
 * the plan.  Per step, patches write inputs[input] + delta into an argument slot (blob < 0) or into 8 bytes of a blob (a pointer field
 * of a descriptor), and every AderDrop.key listed in `keys` is set to the key of (seed, step, site) -- the host half of the dropout
 * counter spec above.  Launchers run in list order; a WAIT op makes lane `stream` wait for everything enqueued on lane `other` so far. */
#define ADER_STEP_MAX_ARGS 48
#define ADER_STEP_MAX_INPUTS 16
enum { ADER_STEP_LAUNCH = 0, ADER_STEP_WAIT = 1 };
typedef struct { int kind, fn, stream, other, n_args, pad_; uint64_t args[ADER_STEP_MAX_ARGS]; } AderStepOp;
typedef struct { int op, arg; const void* src; size_t bytes; } AderStepBlob;
typedef struct { int blob, op, arg, input; size_t offset; int64_t delta; } AderStepPatch;
typedef struct { int blob, site; size_t offset; } AderStepKey;
typedef struct AderStepPlan AderStepPlan;
/* index of a launcher in the plan's dispatch table (-1: not a step launcher) and its argument count */
int ader_step_fn_index(const char* name);
int ader_step_fn_args(int fn);
int ader_step_plan_create(const AderStepOp* ops, int n_ops, const AderStepBlob* blobs, int n_blobs, const AderStepPatch* patches,
                          int n_patches, const AderStepKey* keys, int n_keys, unsigned seed, AderStepPlan** out);
int ader_step_plan_destroy(AderStepPlan* plan);
/* returns 0, or the first failing launcher's / HIP call's code (ader_step_plan_failed_op: which op) */
int ader_step_enqueue(AderStepPlan* plan, const uint64_t* inputs, int n_inputs, unsigned step, void* main_stream, void* side_stream);
/* the patched ops / descriptor bytes a step would issue, without issuing them (blob_out[b]: room for blob b's bytes, or NULL) */
int ader_step_plan_peek(AderStepPlan* plan, const uint64_t* inputs, int n_inputs, unsigned step, AderStepOp* ops_out,
                        void* const* blob_out);
int ader_step_plan_failed_op(const AderStepPlan* plan);

/* ---- cross-check kernels: NOT part of the product library.  Built only into libader_xcheck.so (ader_amd/build.py compiles
 *      table_update.hip and herding.hip a second time with -DADER_XCHECK) and loaded only by the tests that compare kernel against
 *      kernel: the round-2 fused table update on 64-row tiles (x3 with rep_lo != NULL, bf16 "resident" form with rep_lo == NULL) and
 *      the generic herding kernel under its own name. ------------------------------------------------------------------------------- */
#ifdef ADER_XCHECK
/* Fused table update: table-gradient GEMM + sparse terms (input-embedding rows sp_*, one-hot targets tg_*, both sorted by
 * item id) + tf.train.AdamOptimizer (ADER.py:96) on table rows 1..N of emb/adam_m/adam_v, one workgroup per 64-row tile.
 * The table gradient is never written to memory and the item parameters are read ONCE (GEMM operand and Adam input come
 * from the same LDS-resident tile in bf16 mode).  lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t).
 * rep_lo != NULL selects x3 mode.  shadow: bf16 [item_num+1][168] copy of the table, rows rewritten after the update
 * (NULL: none; never written in x3 mode).  extra_grad: dense fp32 gradient [item_num+1, H] (table layout) added row by row
 * before the update, or NULL -- the table gradient of rows that did not go through this path (distilled exemplar rows,
 * ADER.py:132-137).
 * Bucket layout of the sorted lists: bucket j covers ids [g*j + id0, g*(j+1) + id0) with g = ader_fused_bucket_gran(),
 * id0 = ader_fused_bucket_id0(). */
int ader_tab_update(const void* rep_hi, const void* rep_lo, void* shadow, int item_num, int B, int Bp, int H, int N,
                    const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src, float sp_scale,
                    const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta, const float* wrow, float* emb,
                    float* adam_m, float* adam_v, float lr_t, float beta1, float beta2, float eps, int tile_begin,
                    int tile_count, const float* extra_grad, void* stream);
/* ... and for a DISTILLED step at float32 grade (x3; layout and arguments as ader_lx3_fwd_kd left them; ADER.py:132-137) */
int ader_tab_update_kd(const void* rep_hi, const void* rep_lo, int item_num, int Bp, int kd_row0, int H, int N, int Np,
                       const float* off, const int* sp_ids, const int* sp_rows, int n_sp, const float* sp_src, float sp_scale,
                       const int* tg_ids, const int* tg_rows, int n_tg, const int* tile_meta, const float* wrow,
                       const float* teacher, long ldt, const int* trow, const float* tlse2, float* emb, float* adam_m,
                       float* adam_v, float lr_t, float beta1, float beta2, float eps, void* stream);
/* the generic kernel for any H <= 256 (one 256-thread workgroup per group, D streamed from L2 every iteration; D needs only
 * n_total*H floats): what ader_herding_select runs for H != 150, exported for kernel-vs-kernel checks and A/B timing */
int ader_herding_select_generic(const float* rep, const long* seg, const int* quota, const int* max_steps, int G, long n_total,
                                int H, float* D, unsigned char* chosen, int* sel, int* sel_cnt, int* steps_out, void* stream);
#endif /* ADER_XCHECK */

/* ---- host-side feeder helper (no device work, no stream) --------------------------------------------------------------------
 * Python's random.shuffle on an int64 array given the `random` module's Mersenne-Twister state (mt_state[0..623] words, [624] index =
 * random.getstate()[1]); array and state are advanced in place exactly as CPython advances them.  The reference's Sampler
 * re-shuffles its index list every epoch (util.py:152-157, 226-235) and every later draw of the run continues that stream. */
int ader_host_shuffle(uint32_t* mt_state, int64_t* x, int64_t n);
/* The Sampler's packed rows (util.py:161-169, 226-227) of n sessions given as one flat int32 item array + lens [n]: rows [n][maxlen+1]
 * (ZERO on entry) = up to the last maxlen inputs right-aligned, then the label; valid [n] bytes = session has at least 2 items. */
int ader_host_pack_rows(const int32_t* flat, const int64_t* lens, int64_t n, int maxlen, int32_t* rows, unsigned char* valid);
/* ... of n sessions given as (starts[i], lens[i]) into a shared flat item array of flat_n items (a prefix of a session = the same start
 * with a shorter length; a split = a gather of pairs): the array data plane of ader_amd/data.py (PackedSessions). */
int ader_host_pack_rows_at(const int32_t* flat, int64_t flat_n, const int64_t* starts, const int64_t* lens, int64_t n, int maxlen,
                           int32_t* rows, unsigned char* valid);
/* ... of every session and its prefixes down to length 2 (util.py:138-143), in the Sampler's order, without building the prefix lists:
 * rows [sum_i max(1, lens[i] - 1)][maxlen+1] (ZERO on entry), valid one byte per row. */
int ader_host_prefix_rows(const int32_t* flat, const int64_t* lens, int64_t n, int maxlen, int32_t* rows, unsigned char* valid);

#ifdef __cplusplus
}
#endif
#endif
