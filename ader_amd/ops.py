"""`torch.ops.ader.*`: the hot-path launchers of libader_hip.so as PyTorch custom operators (SURVEY 8b "Native ABI": the north star
asks for the kernels "surfaced to Python via PyTorch-ROCm custom ops").

The C ABI (include/ader_hip.h) stays the drop-in boundary; these operators are the torch-facing view of the same launchers for
code that wants to compose them with other torch ops: each takes and returns device tensors, allocates its outputs with the
caching allocator, launches on torch's CURRENT stream, checks dtype / device / contiguity / shape and raises RuntimeError on a
violation (never clamps ids, never falls back to the CPU).  `import ader_amd.ops` registers them.

  ader::embed_fwd(seq, emb, pos, key, thr, scale)            -> x      ADER.py:41-60, modules.py:118-130
  ader::layernorm_fwd(x, gamma, beta)                          -> (y, mean, std)   modules.py:23-50
  ader::logits_ce_fwd(rep, shadow, labels, weights, N)         -> (loss, lse, drep, off)   ADER.py:88-93 (bf16 flash forward)
  ader::adam_step(p, m, v, g, lr_t, beta1, beta2, eps)         -> ()     in place, ADER.py:96 (TF ApplyAdam)
  ader::rank_of_target(rep, emb, target, N)                    -> rank   ADER.py:99-103 + util.py:325
  ader::herding_select(rep, seg, quota, max_steps)             -> (sel, cnt)   util.py:401-434

Trainable surface (second half of this file; every forward has an autograd formula over its `_bwd` operator):
  ader::embed_fwd / embed_bwd                                  ADER.py:25-60, modules.py:118-130
  ader::layernorm / layernorm_bwd                              modules.py:23-50 (+ the key / query mask bits)
  ader::attn_fwd / attn_bwd                                    modules.py:135-229 (Q/K/V projections included)
  ader::ffn_fwd / ffn_bwd                                      modules.py:232-271 + ADER.py:80
  ader::logits_ce / logits_ce_bwd                              ADER.py:88-93 (exact-f32 logits, one-hot CE)
"""

import os

import torch

from . import _lib
from ._lib import call, ptr

_lib.load()


def _chk(t, name, dtype, dims=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("ader::%s must be a CUDA/HIP tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("ader::%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("ader::%s must be contiguous" % name)
    if dims is not None and t.dim() != dims:
        raise RuntimeError("ader::%s must have %d dimensions" % (name, dims))


def _st():
    return torch.cuda.current_stream().cuda_stream


@torch.library.custom_op("ader::embed_fwd", mutates_args=())
def embed_fwd(seq: torch.Tensor, emb: torch.Tensor, pos: torch.Tensor, key: int, thr: int, scale: float) -> torch.Tensor:
    _chk(seq, "seq", torch.int32, 2), _chk(emb, "emb", torch.float32, 2), _chk(pos, "pos", torch.float32, 2)
    B, T = seq.shape
    V, H = emb.shape
    if pos.shape != (T, H):
        raise RuntimeError("ader::embed_fwd: pos must be [T, H]")
    x = torch.empty(B * T, H, dtype=torch.float32, device=seq.device)
    status = torch.zeros(1, dtype=torch.int32, device=seq.device)
    import ctypes
    d = _lib.AderDrop(key & 0xFFFFFFFF, thr, scale, 0, 0xFFFFFFFF, 0)
    call("ader_embed_fwd", ptr(seq), ptr(emb), ptr(pos), ptr(x), B * T, T, H, V, ctypes.byref(d), ptr(status), _st())
    if int(status.item()):
        raise RuntimeError("ader::embed_fwd: item id outside [0, V) in seq")
    return x.view(B, T, H)


@embed_fwd.register_fake
def _(seq, emb, pos, key, thr, scale):
    return emb.new_empty(seq.shape[0], seq.shape[1], emb.shape[1])


@torch.library.custom_op("ader::layernorm_fwd", mutates_args=())
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    _chk(x, "x", torch.float32, 2), _chk(gamma, "gamma", torch.float32, 1), _chk(beta, "beta", torch.float32, 1)
    rows, H = x.shape
    if gamma.shape[0] != H or beta.shape[0] != H:
        raise RuntimeError("ader::layernorm_fwd: gamma / beta must be [H]")
    y = torch.empty_like(x)
    mean, std = torch.empty(rows, device=x.device), torch.empty(rows, device=x.device)
    call("ader_ln_fwd", ptr(x), H, ptr(y), H, ptr(gamma), ptr(beta), ptr(mean), ptr(std), None, None, rows, H, _st())
    return y, mean, std


@layernorm_fwd.register_fake
def _(x, gamma, beta):
    return torch.empty_like(x), x.new_empty(x.shape[0]), x.new_empty(x.shape[0])


@torch.library.custom_op("ader::logits_ce_fwd", mutates_args=())
def logits_ce_fwd(rep: torch.Tensor, shadow: torch.Tensor, labels: torch.Tensor, weights: torch.Tensor,
                  N: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """rep [B,H] fp32; shadow [V*168] bf16 (ader_lbf_shadow_refresh layout); labels int32 [B] (1-based item ids); weights fp32 [B]
    (1/B for the mean of ADER.py:93).  Returns (loss [1], lse [B], drep [B,H], off [Bp])."""
    _chk(rep, "rep", torch.float32, 2), _chk(shadow, "shadow", torch.bfloat16, 1)
    _chk(labels, "labels", torch.int32, 1), _chk(weights, "weights", torch.float32, 1)
    B, H = rep.shape
    item_num = shadow.numel() // 168 - 1
    if not (1 <= N <= item_num) or labels.shape[0] != B or weights.shape[0] != B or H % 2:
        raise RuntimeError("ader::logits_ce_fwd: bad shapes (N=%d, item_num=%d, B=%d)" % (N, item_num, B))
    _check_labels(labels, "logits_ce_fwd labels", N)
    Bp = (B + 127) // 128 * 128
    dev = rep.device
    lab, w = torch.zeros(Bp, dtype=torch.int32, device=dev), torch.zeros(Bp, device=dev)
    lab[:B], w[:B] = labels, weights
    R = call("ader_lbf_ranges", N, Bp)
    rep_bf = torch.empty(Bp * 168, dtype=torch.bfloat16, device=dev)
    pm, pl, pO = torch.empty(R * Bp, device=dev), torch.empty(R * Bp, device=dev), torch.empty(R * Bp * 160, device=dev)
    lse, off, rowloss = torch.empty(Bp, device=dev), torch.empty(Bp, device=dev), torch.empty(Bp, device=dev)
    loss, drep = torch.zeros(1, device=dev), torch.empty(B, H, device=dev)
    call("ader_lbf_fwd", ptr(rep), ptr(shadow), item_num, B, Bp, H, N, ptr(lab), ptr(w), ptr(rep_bf), ptr(pm), ptr(pl), ptr(pO),
         ptr(lse), ptr(off), ptr(rowloss), ptr(loss), ptr(drep), _st())
    return loss, lse[:B].clone(), drep, off


@logits_ce_fwd.register_fake
def _(rep, shadow, labels, weights, N):
    B = rep.shape[0]
    return rep.new_empty(1), rep.new_empty(B), torch.empty_like(rep), rep.new_empty((B + 127) // 128 * 128)


@torch.library.custom_op("ader::adam_step", mutates_args=("p", "m", "v"))
def adam_step(p: torch.Tensor, m: torch.Tensor, v: torch.Tensor, g: torch.Tensor, lr_t: float, beta1: float, beta2: float,
              eps: float) -> None:
    for t, n in ((p, "p"), (m, "m"), (v, "v"), (g, "g")):
        _chk(t, n, torch.float32, 1)
        if t.numel() != p.numel():
            raise RuntimeError("ader::adam_step: p, m, v, g must have the same length")
    call("ader_adam_step", ptr(p), ptr(m), ptr(v), ptr(g), p.numel(), lr_t, beta1, beta2, eps, None, 0, 1, _st())


@torch.library.custom_op("ader::rank_of_target", mutates_args=())
def rank_of_target(rep: torch.Tensor, emb: torch.Tensor, target: torch.Tensor, N: int) -> torch.Tensor:
    _chk(rep, "rep", torch.float32, 2), _chk(emb, "emb", torch.float32, 2), _chk(target, "target", torch.int32, 1)
    B, H = rep.shape
    if emb.shape[1] != H or not (1 <= N <= emb.shape[0] - 1) or target.shape[0] != B or B > 1024:
        raise RuntimeError("ader::rank_of_target: bad shapes")
    Bp = (B + 63) // 64 * 64
    dev = rep.device
    tgt, ncol = torch.zeros(Bp, dtype=torch.int32, device=dev), torch.zeros(Bp, dtype=torch.int32, device=dev)
    tgt[:B], ncol[:B] = target, N
    tl, rk = torch.empty(Bp, device=dev), torch.empty(Bp, dtype=torch.int32, device=dev)
    call("ader_rank_targets", ptr(rep), ptr(emb), B, Bp, H, N, ptr(tgt), ptr(ncol), ptr(tl), ptr(rk), _st())
    return rk[:B].clone()


@rank_of_target.register_fake
def _(rep, emb, target, N):
    return target.new_empty(rep.shape[0])


@torch.library.custom_op("ader::herding_select", mutates_args=())
def herding_select(rep: torch.Tensor, seg: torch.Tensor, quota: torch.Tensor, max_steps: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """rep [n,H] fp32 candidates in group order; seg int64 [G+1] group offsets; quota / max_steps int32 [G].  Returns (sel int32 [n]:
    per group, the selected LOCAL indices in selection order at the start of its span; cnt int32 [G])."""
    _chk(rep, "rep", torch.float32, 2), _chk(seg, "seg", torch.int64, 1), _chk(quota, "quota", torch.int32, 1)
    _chk(max_steps, "max_steps", torch.int32, 1)
    n, H = rep.shape
    G = quota.shape[0]
    if seg.shape[0] != G + 1 or max_steps.shape[0] != G:
        raise RuntimeError("ader::herding_select: seg must be [G+1], quota / max_steps [G]")
    dev = rep.device
    D = torch.empty(n * H + G + 64, device=dev)
    chosen = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    sel = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(max(G, 1), dtype=torch.int32, device=dev)
    call("ader_herding_select", ptr(rep), ptr(seg), ptr(quota), ptr(max_steps), G, n, H, ptr(D), ptr(chosen), ptr(sel), ptr(cnt), None,
         _st())
    return sel[:n], cnt[:G]


@herding_select.register_fake
def _(rep, seg, quota, max_steps):
    return quota.new_empty(rep.shape[0]), quota.new_empty(quota.shape[0])


# ================================================================================================ trainable surface
# SURVEY 8(b) "Native ABI": embed_fwd/bwd, layernorm_fwd/bwd, attn_fwd/bwd, ffn_fwd/bwd, logits_ce_fwd/bwd over the exact-f32
# launchers, each forward with a registered autograd formula, so that torch.ops.ader.* composes into a trainable graph
# (tests/test_gpu_ops_autograd.py trains one SASRec step through these ops + torch.autograd and compares every gradient with
# Engine.loss_and_grad).  Dropout is the counter-hash of the kernels: (key, thr, scale) per site, thr = 0 disables it.
import ctypes


def _drop(key, thr, scale):
    return ctypes.byref(_lib.AderDrop(key & 0xFFFFFFFF, thr, scale, 0, 0xFFFFFFFF, 0))


def _atb(A, G):
    """dW = A^T G [H,H], db = colsum(G) [H]"""
    M, H = A.shape
    slab = torch.empty(call("ader_gemm_atb_slabs", M) * 160 * 160, device=A.device)
    dW, db = torch.empty(H, H, device=A.device), torch.empty(H, device=A.device)
    call("ader_gemm_atb", ptr(A), ptr(G), ptr(slab), ptr(dW), ptr(db), M, H, _st())
    return dW, db


def _gemm(A, W, bias, aux, seq, epi, trans, drop=None):
    M, H = A.shape
    C = torch.empty(M, H, device=A.device)
    call("ader_gemm_rows", ptr(A), ptr(W), ptr(bias), ptr(C), ptr(aux), ptr(seq), M, H, epi, trans, 1, 0, drop, _st())
    return C


# ---- embedding prologue: x = drop(E0[seq] * sqrt(H) + P) * (seq != 0)   (ADER.py:25-60, modules.py:118-130)
@torch.library.custom_op("ader::embed_bwd", mutates_args=())
def embed_bwd(seq: torch.Tensor, dx: torch.Tensor, V: int, key: int, thr: int, scale: float) -> tuple[torch.Tensor, torch.Tensor]:
    _chk(seq, "seq", torch.int32, 2), _chk(dx, "dx", torch.float32, 3)
    B, T = seq.shape
    H = dx.shape[2]
    g = dx.clone()                                   # (the launcher scales the rows in place)
    dev = dx.device
    demb, dpos = torch.zeros(V, H, device=dev), torch.zeros(T, H, device=dev)
    # rows first (mask, dropout, positional gradient), then a bucketed, position-ordered accumulation into the table gradient: no
    # float atomics, bitwise reproducible
    call("ader_embed_bwd_rows", ptr(seq), ptr(g), ptr(dpos), B, T, H, V, _drop(key, thr, scale), _st())
    N = V - 1
    i32 = dict(dtype=torch.int32, device=dev)
    n_sp = B * T
    lab0 = torch.zeros(1, **i32)
    nb1 = call("ader_sparse_lists_starts", N)
    ids, order = torch.empty(n_sp, **i32), torch.empty(n_sp, **i32)
    tids, torder = torch.empty(1, **i32), torch.empty(1, **i32)
    sp_start, tg_start = torch.empty(nb1, **i32), torch.empty(nb1, **i32)
    scratch = torch.empty(call("ader_sparse_lists_scratch_n", n_sp, 1, N), **i32)
    call("ader_sparse_lists", ptr(seq), n_sp, ptr(lab0), 1, N, ptr(scratch), ptr(ids), ptr(order), ptr(sp_start), ptr(tids), ptr(torder),
         ptr(tg_start), _st())
    call("ader_scatter_rows_ordered", ptr(ids), ptr(order), ptr(sp_start), nb1 - 1, ptr(g), H, V, _sqrt_f32(H), ptr(demb), _st())
    return demb, dpos


@embed_bwd.register_fake
def _(seq, dx, V, key, thr, scale):
    return dx.new_empty(V, dx.shape[2]), dx.new_empty(seq.shape[1], dx.shape[2])


def _embed_setup(ctx, inputs, output):
    seq, emb, pos, key, thr, scale = inputs
    ctx.save_for_backward(seq)
    ctx.meta = (emb.shape[0], key, thr, scale)


def _embed_backward(ctx, dx):
    (seq,) = ctx.saved_tensors
    V, key, thr, scale = ctx.meta
    demb, dpos = torch.ops.ader.embed_bwd(seq, dx.contiguous(), V, key, thr, scale)
    return None, demb, dpos, None, None, None


torch.library.register_autograd("ader::embed_fwd", _embed_backward, setup_context=_embed_setup)


# ---- LayerNorm (modules.py:23-50) with the key / query masks of modules.py:188-193, 208-211 as by-products
@torch.library.custom_op("ader::layernorm", mutates_args=())
def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (y, mean, std, xnz, ynz): xnz[r] = (sum_c x[r,c] != 0) (key mask), ynz[r] = (sum_c y[r,c] != 0) (query mask)"""
    _chk(x, "x", torch.float32, 2), _chk(gamma, "gamma", torch.float32, 1), _chk(beta, "beta", torch.float32, 1)
    rows, H = x.shape
    if gamma.shape[0] != H or beta.shape[0] != H:
        raise RuntimeError("ader::layernorm: gamma / beta must be [H]")
    y = torch.empty_like(x)
    mean, std, xnz, ynz = (torch.empty(rows, device=x.device) for _ in range(4))
    call("ader_ln_fwd", ptr(x), H, ptr(y), H, ptr(gamma), ptr(beta), ptr(mean), ptr(std), ptr(xnz), ptr(ynz), rows, H, _st())
    return y, mean, std, xnz, ynz


@layernorm.register_fake
def _(x, gamma, beta):
    r = x.shape[0]
    return torch.empty_like(x), x.new_empty(r), x.new_empty(r), x.new_empty(r), x.new_empty(r)


@torch.library.custom_op("ader::layernorm_bwd", mutates_args=())
def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor,
                  std: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    _chk(dy, "dy", torch.float32, 2), _chk(x, "x", torch.float32, 2)
    rows, H = x.shape
    dx = torch.empty_like(x)
    slab = torch.empty(call("ader_ln_bwd_slabs", rows) * 2 * H, device=x.device)
    dg, db = torch.empty(H, device=x.device), torch.empty(H, device=x.device)
    call("ader_ln_bwd", ptr(dy), H, ptr(x), H, ptr(gamma), ptr(mean), ptr(std), None, 0, ptr(dx), H, ptr(slab), ptr(dg), ptr(db),
         rows, H, _st())
    return dx, dg, db


@layernorm_bwd.register_fake
def _(dy, x, gamma, mean, std):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


def _ln_setup(ctx, inputs, output):
    x, gamma, beta = inputs
    ctx.save_for_backward(x, gamma, output[1], output[2])


def _ln_backward(ctx, dy, dmean, dstd, dxnz, dynz):
    x, gamma, mean, std = ctx.saved_tensors
    dx, dg, db = torch.ops.ader.layernorm_bwd(dy.contiguous(), x, gamma, mean, std)
    return dx, dg, db


torch.library.register_autograd("ader::layernorm", _ln_backward, setup_context=_ln_setup)


# ---- multi-head attention of modules.py:135-229: Q = q_in Wq + bq, K = x Wk + bk, V = x Wv + bv, causal + key masks, softmax,
# query mask, dropout on the probabilities, . V, + q_in (the NORMALISED queries as residual)
@torch.library.custom_op("ader::attn_fwd", mutates_args=())
def attn_fwd(x: torch.Tensor, q_in: torch.Tensor, wq: torch.Tensor, bq: torch.Tensor, wk: torch.Tensor, bk: torch.Tensor,
             wv: torch.Tensor, bv: torch.Tensor, kmask: torch.Tensor, qmask: torch.Tensor, B: int, T: int, heads: int, key: int,
             thr: int, scale: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """x, q_in [B*T,H] -> (out [B*T,H], Q, K, V, P [B*heads*T*T])"""
    _chk(x, "x", torch.float32, 2), _chk(q_in, "q_in", torch.float32, 2)
    rows, H = x.shape
    if rows != B * T or q_in.shape != x.shape or H % heads:
        raise RuntimeError("ader::attn_fwd: x, q_in must be [B*T, H] with H % heads == 0")
    Q = _gemm(q_in, wq, bq, None, None, 0, 0)
    K = _gemm(x, wk, bk, None, None, 0, 0)
    Vv = _gemm(x, wv, bv, None, None, 0, 0)
    out = torch.empty_like(x)
    P = torch.empty(B * heads * T * T, device=x.device)
    call("ader_attn_fwd", ptr(Q), ptr(K), ptr(Vv), ptr(q_in), ptr(kmask), ptr(qmask), ptr(out), ptr(P), B, T, H, heads,
         _drop(key, thr, scale), _st())
    return out, Q, K, Vv, P


@attn_fwd.register_fake
def _(x, q_in, wq, bq, wk, bk, wv, bv, kmask, qmask, B, T, heads, key, thr, scale):
    return torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), x.new_empty(B * heads * T * T)


@torch.library.custom_op("ader::attn_bwd", mutates_args=())
def attn_bwd(dO: torch.Tensor, x: torch.Tensor, q_in: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor,
             Q: torch.Tensor, K: torch.Tensor, Vv: torch.Tensor, P: torch.Tensor, kmask: torch.Tensor, qmask: torch.Tensor, B: int,
             T: int, heads: int, key: int, thr: int,
             scale: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (dx, dq_in, dwq, dbq, dwk, dbk, dwv, dbv)"""
    rows, H = x.shape
    dQ, dK, dV = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    call("ader_attn_bwd", ptr(dO), ptr(Q), ptr(K), ptr(Vv), ptr(P), ptr(kmask), ptr(qmask), ptr(dQ), ptr(dK), ptr(dV), B, T, H, heads,
         _drop(key, thr, scale), _st())
    dq_in = _gemm(dQ, wq, None, dO, None, 4, 1)                       # dQ Wq^T + dO (the residual adds the normalised queries)
    dx = _gemm(dK, wk, None, None, None, 0, 1)
    dx = _gemm(dV, wv, None, dx, None, 4, 1)
    dwq, dbq = _atb(q_in, dQ)
    dwk, dbk = _atb(x, dK)
    dwv, dbv = _atb(x, dV)
    return dx, dq_in, dwq, dbq, dwk, dbk, dwv, dbv


@attn_bwd.register_fake
def _(dO, x, q_in, wq, wk, wv, Q, K, Vv, P, kmask, qmask, B, T, heads, key, thr, scale):
    e = torch.empty_like
    return e(x), e(x), e(wq), wq.new_empty(wq.shape[0]), e(wk), wk.new_empty(wk.shape[0]), e(wv), wv.new_empty(wv.shape[0])


def _attn_setup(ctx, inputs, output):
    x, q_in, wq, bq, wk, bk, wv, bv, kmask, qmask, B, T, heads, key, thr, scale = inputs
    ctx.save_for_backward(x, q_in, wq, wk, wv, output[1], output[2], output[3], output[4], kmask, qmask)
    ctx.meta = (B, T, heads, key, thr, scale)


def _attn_backward(ctx, dO, dQ_, dK_, dV_, dP_):
    x, q_in, wq, wk, wv, Q, K, Vv, P, kmask, qmask = ctx.saved_tensors
    dx, dq_in, dwq, dbq, dwk, dbk, dwv, dbv = torch.ops.ader.attn_bwd(dO.contiguous(), x, q_in, wq, wk, wv, Q, K, Vv, P, kmask, qmask,
                                                                      *ctx.meta)
    return dx, dq_in, dwq, dbq, dwk, dbk, dwv, dbv, None, None, None, None, None, None, None, None


torch.library.register_autograd("ader::attn_fwd", _attn_backward, setup_context=_attn_setup)


# ---- position-wise feed-forward of modules.py:232-271 + ADER.py:80: (drop(drop(relu(y W1 + b1)) W2 + b2) + y) * (seq != 0)
@torch.library.custom_op("ader::ffn_fwd", mutates_args=())
def ffn_fwd(y: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, seq: torch.Tensor, key1: int,
            thr1: int, scale1: float, key2: int, thr2: int, scale2: float) -> tuple[torch.Tensor, torch.Tensor]:
    """y [B*T,H] (the LayerNorm'd block input), seq int32 [B,T] -> (x2, h1d = drop(relu(y W1 + b1)))"""
    _chk(y, "y", torch.float32, 2), _chk(seq, "seq", torch.int32, 2)
    if y.shape[0] != seq.numel():
        raise RuntimeError("ader::ffn_fwd: y must have one row per position of seq")
    h1d = _gemm(y, w1, b1, None, None, 1, 0, _drop(key1, thr1, scale1))
    x2 = _gemm(h1d, w2, b2, y, seq, 2, 0, _drop(key2, thr2, scale2))
    return x2, h1d


@ffn_fwd.register_fake
def _(y, w1, b1, w2, b2, seq, key1, thr1, scale1, key2, thr2, scale2):
    return torch.empty_like(y), torch.empty_like(y)


@torch.library.custom_op("ader::ffn_bwd", mutates_args=())
def ffn_bwd(dx2: torch.Tensor, y: torch.Tensor, h1d: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, seq: torch.Tensor, key1: int,
            thr1: int, scale1: float, key2: int, thr2: int,
            scale2: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (dy, dw1, db1, dw2, db2)"""
    M, H = y.shape
    g, dh2 = torch.empty_like(y), torch.empty_like(y)
    call("ader_mask_dropgrad", ptr(dx2), ptr(seq), ptr(g), ptr(dh2), M, H, 1, 0, _drop(key2, thr2, scale2), _st())
    da = _gemm(dh2, w2, None, h1d, None, 3, 1, _drop(key1, thr1, scale1))          # through W2^T, the first dropout and the ReLU
    dy = _gemm(da, w1, None, g, None, 4, 1)                                        # through W1^T, + the residual's gradient
    dw2, db2 = _atb(h1d, dh2)
    dw1, db1 = _atb(y, da)
    return dy, dw1, db1, dw2, db2


@ffn_bwd.register_fake
def _(dx2, y, h1d, w1, w2, seq, key1, thr1, scale1, key2, thr2, scale2):
    e = torch.empty_like
    return e(y), e(w1), w1.new_empty(w1.shape[0]), e(w2), w2.new_empty(w2.shape[0])


def _ffn_setup(ctx, inputs, output):
    y, w1, b1, w2, b2, seq, *d = inputs
    ctx.save_for_backward(y, output[1], w1, w2, seq)
    ctx.meta = tuple(d)


def _ffn_backward(ctx, dx2, dh1d):
    y, h1d, w1, w2, seq = ctx.saved_tensors
    dy, dw1, db1, dw2, db2 = torch.ops.ader.ffn_bwd(dx2.contiguous(), y, h1d, w1, w2, seq, *ctx.meta)
    return dy, dw1, db1, dw2, db2, None, None, None, None, None, None, None


torch.library.register_autograd("ader::ffn_fwd", _ffn_backward, setup_context=_ffn_setup)


# ---- full-catalog logits + one-hot softmax cross entropy (ADER.py:88-93), exact-f32 kernels: loss = sum_b w_b (lse_b - s_b[label_b])
def _rowinfo(labels, weights, N, dev):
    B = labels.shape[0]
    Bp = (B + 63) // 64 * 64
    lab, ncol = torch.zeros(Bp, dtype=torch.int32, device=dev), torch.zeros(Bp, dtype=torch.int32, device=dev)
    w, trow, tlse = torch.zeros(Bp, device=dev), torch.full((Bp,), -1, dtype=torch.int32, device=dev), torch.zeros(Bp, device=dev)
    lab[:B], ncol[:B], w[:B] = labels, N, weights
    return Bp, lab, ncol, w, trow, tlse


@torch.library.custom_op("ader::logits_ce", mutates_args=())
def logits_ce(rep: torch.Tensor, emb: torch.Tensor, labels: torch.Tensor, weights: torch.Tensor, N: int) -> tuple[torch.Tensor, torch.Tensor]:
    """rep [B,H], emb [V,H] (row 0 = padding item), labels int32 [B] 1-based, weights [B] -> (loss [1], lse [B])"""
    _chk(rep, "rep", torch.float32, 2), _chk(emb, "emb", torch.float32, 2), _chk(labels, "labels", torch.int32, 1)
    _chk(weights, "weights", torch.float32, 1)
    B, H = rep.shape
    if emb.shape[1] != H or not (1 <= N <= emb.shape[0] - 1) or labels.shape[0] != B or weights.shape[0] != B or B > 1024:
        raise RuntimeError("ader::logits_ce: bad shapes")
    _check_labels(labels, "logits_ce labels", N)
    dev = rep.device
    Bp, lab, ncol, w, trow, tlse = _rowinfo(labels, weights, N, dev)
    part = torch.empty(call("ader_logits_parts", N) * Bp * 3, device=dev)
    lse, rowloss, loss = torch.empty(Bp, device=dev), torch.empty(Bp, device=dev), torch.zeros(1, device=dev)
    call("ader_logits_loss_fwd", ptr(rep), ptr(emb), B, Bp, H, N, ptr(lab), ptr(ncol), ptr(w), ptr(trow), ptr(tlse), None, 0,
         ptr(part), ptr(lse), ptr(rowloss), ptr(loss), _st())
    return loss, lse[:B].clone()


@logits_ce.register_fake
def _(rep, emb, labels, weights, N):
    return rep.new_empty(1), rep.new_empty(rep.shape[0])


@torch.library.custom_op("ader::logits_ce_bwd", mutates_args=())
def logits_ce_bwd(rep: torch.Tensor, emb: torch.Tensor, labels: torch.Tensor, weights: torch.Tensor, lse: torch.Tensor,
                  N: int) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (drep [B,H], demb [V,H]) for d loss = 1"""
    _chk(rep, "rep", torch.float32, 2), _chk(emb, "emb", torch.float32, 2), _chk(labels, "labels", torch.int32, 1)
    _chk(weights, "weights", torch.float32, 1), _chk(lse, "lse", torch.float32, 1)
    B, H = rep.shape
    if emb.shape[1] != H or not (1 <= N <= emb.shape[0] - 1) or labels.shape[0] != B or weights.shape[0] != B or lse.shape[0] != B or B > 1024:
        raise RuntimeError("ader::logits_ce_bwd: bad shapes")
    _check_labels(labels, "logits_ce_bwd labels", N)
    dev = rep.device
    Bp, lab, ncol, w, trow, tlse = _rowinfo(labels, weights, N, dev)
    lse_p = torch.zeros(Bp, device=dev)
    lse_p[:B] = lse
    slab = torch.empty(call("ader_logits_ranges", N, Bp) * Bp * 160, device=dev)
    drep, demb = torch.empty(B, H, device=dev), torch.zeros_like(emb)
    call("ader_logits_bwd_drep", ptr(rep), ptr(emb), B, Bp, H, N, ptr(lab), ptr(ncol), ptr(w), ptr(trow), ptr(tlse), None, 0,
         ptr(lse_p), ptr(slab), ptr(drep), _st())
    call("ader_logits_bwd_demb", ptr(rep), ptr(emb), B, Bp, H, N, ptr(lab), ptr(ncol), ptr(w), ptr(trow), ptr(tlse), None, 0,
         ptr(lse_p), ptr(demb), _st())
    return drep, demb


@logits_ce_bwd.register_fake
def _(rep, emb, labels, weights, lse, N):
    return torch.empty_like(rep), torch.empty_like(emb)


def _lce_setup(ctx, inputs, output):
    rep, emb, labels, weights, N = inputs
    ctx.save_for_backward(rep, emb, labels, weights, output[1])
    ctx.N = N


def _lce_backward(ctx, dloss, dlse):
    rep, emb, labels, weights, lse = ctx.saved_tensors
    drep, demb = torch.ops.ader.logits_ce_bwd(rep, emb, labels, weights, lse, ctx.N)
    return drep * dloss, demb * dloss, None, None, None


torch.library.register_autograd("ader::logits_ce", _lce_backward, setup_context=_lce_setup)


# ================================================================================================ the float32-grade FAST path
# The headline kernels as operators (north star: "surfaced to Python via PyTorch-ROCm custom ops"): the flash logit forward that
# yields loss, log-sum-exp and dRep in one pass over the catalog (k_lx3p), its distilled variant (ADER.py:132-137), and the fused
# table-gradient + dense TF-Adam update (k_tab32x3) as the optimizer-step op of the item table.  The [B, N] logits and the [N, H] table
# gradient never exist in memory, so `emb` gets no autograd gradient from these ops: the table is trained by table_update_x3.
#   ader::logits_ce_x3(rep, emb, pos, ex_pos, N, w_train, w_ex)            -> (loss, lse, drep, rep_hi, rep_lo, off, lab, wrow, img)
#   ader::logits_ce_x3_kd(rep, emb, pos, ex_trow, teacher, N, w_train, w_ex) -> (... , trow, tlse2)
#   ader::table_update_x3(emb, m, v, seq, g_rows, rep_hi, rep_lo, off, lab, wrow, img, n_rows, N, lr_t, b1, b2, eps)
#   ader::table_update_x3_kd(emb, m, v, seq, g_rows, rep_hi, rep_lo, off, lab, wrow, teacher, trow, tlse2, n_train, N, lr_t, ...)
# tests/test_gpu_ops_autograd.py steps the table through them and compares with Engine.train_step bit for bit.
def _sqrt_f32(H):
    import numpy as np
    return float(np.sqrt(np.float32(H)))        # the engine's float32 sqrt(H) (ADER.py:38 scale), not the double one


_STATUS = {}        # device -> int32[1]: ids outside their range seen by an op since the last check_status()
# strict (default): an out-of-range label raises at the call that passed it (one host synchronisation per logits_ce call: these ops are
# the autograd surface, not the engine's hot loop).  ader_amd.ops.STRICT_LABELS = False (or ADER_OPS_STRICT=0) keeps the ops free of
# host synchronisation -- stream-ordered, capturable -- and defers the report to check_status().
STRICT_LABELS = os.environ.get("ADER_OPS_STRICT", "1") != "0"


def _check_labels(t, name, N, lo=0):
    """ids the kernels match by equality: an out-of-range label would silently drop its target term (the loss degrades to lse).
    Strict mode raises here.  Otherwise it is flagged ON THE DEVICE (no host synchronisation) into a status word that
    ader_amd.ops.check_status() reads -- as Engine.check_status does for item ids.  0 is legal: "no target" (a weight-0 padding
    row of an equal-size data-parallel shard)."""
    if t.numel():
        bad = ((t < lo) | (t > N)).any()
        if STRICT_LABELS:
            if bool(bad.item()):
                raise RuntimeError("ader::ops: %s holds an id outside [%d, %d]" % (name, lo, N))
            return
        st = _STATUS.get(t.device)
        if st is None:
            st = _STATUS[t.device] = torch.zeros(1, dtype=torch.int32, device=t.device)
        st.bitwise_or_(bad.to(torch.int32))


def check_status():
    """Raise if an op of this module has seen an id outside its range since the last call (one host synchronisation per device that
    ran an op).  Every device's flag is read and cleared before anything is raised."""
    bad = []
    for dev, st in _STATUS.items():
        if int(st.item()):
            bad.append(str(dev))
        st.zero_()
    if bad:
        raise RuntimeError("ader::ops: a label / row id outside its valid range was passed to a logits_ce op on %s" % ", ".join(bad))


def _x3_fwd_scratch(N, Bp, dev):
    R = call("ader_lbf_ranges", N, Bp)
    bf = dict(dtype=torch.bfloat16, device=dev)
    return (torch.empty(Bp * 168, **bf), torch.empty(Bp * 168, **bf), torch.empty(R * Bp, device=dev), torch.empty(R * Bp, device=dev),
            torch.empty(R * Bp * 160, device=dev), torch.empty(Bp, device=dev), torch.empty(Bp, device=dev), torch.empty(Bp, device=dev))


@torch.library.custom_op("ader::logits_ce_x3", mutates_args=())
def logits_ce_x3(rep: torch.Tensor, emb: torch.Tensor, pos: torch.Tensor, ex_pos: torch.Tensor, N: int, w_train: float,
                 w_ex: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
                                       torch.Tensor, torch.Tensor]:
    """One-hot softmax cross entropy over the whole catalog at float32 grade (ADER.py:88-93; exemplar rows with one-hot labels:
    ADER.py:126-131).  rep [B,H] fp32 with the train rows first; emb [V,H] fp32 (row 0 = padding item); pos int32 [n_train], ex_pos
    int32 [B - n_train] (may be empty), 1-based labels; loss = w_train sum_train CE + w_ex sum_ex CE.  Returns loss [1], lse [B],
    drep [B,H] (= d loss / d rep) and what table_update_x3 consumes: operand planes rep_hi / rep_lo [Bp*168] bf16, exponent offsets
    off [Bp], labels lab [Bp], weights wrow [Bp], operand image img (uint8)."""
    _chk(rep, "rep", torch.float32, 2), _chk(emb, "emb", torch.float32, 2), _chk(pos, "pos", torch.int32, 1)
    _chk(ex_pos, "ex_pos", torch.int32, 1)
    B, H = rep.shape
    n_train, n_ex = pos.shape[0], ex_pos.shape[0]
    item_num = emb.shape[0] - 1
    if emb.shape[1] != H or H % 2 or H > 150 or not (1 <= N <= item_num) or n_train + n_ex != B or B > 4096:
        raise RuntimeError("ader::logits_ce_x3: bad shapes (B=%d, n_train=%d, n_ex=%d, H=%d, N=%d, item_num=%d)"
                           % (B, n_train, n_ex, H, N, item_num))
    _check_labels(pos, "logits_ce_x3 pos", N), _check_labels(ex_pos, "logits_ce_x3 ex_pos", N)
    dev = rep.device
    Bp = (B + 127) // 128 * 128
    i32 = dict(dtype=torch.int32, device=dev)
    lab, ncol, trow = torch.empty(Bp, **i32), torch.empty(Bp, **i32), torch.empty(Bp, **i32)
    wrow = torch.empty(Bp, device=dev)
    call("ader_build_rowinfo", ptr(pos), n_train, ptr(ex_pos) if n_ex else None, None, n_ex, N, 0, float(w_train), float(w_ex), Bp,
         ptr(lab), ptr(ncol), ptr(wrow), ptr(trow), _st())
    rep_hi, rep_lo, pm, pl, pO, lse, off, rowloss = _x3_fwd_scratch(N, Bp, dev)
    img = torch.zeros(call("ader_x3_rep_image_bytes", Bp), dtype=torch.uint8, device=dev)
    loss, drep = torch.zeros(1, device=dev), torch.empty(B, H, device=dev)
    # (loss = NULL + ader_lbf_sum: the summation order Engine.train_step uses, so the two agree bit for bit)
    call("ader_lx3_fwd_img", ptr(rep), ptr(emb), item_num, B, Bp, H, N, ptr(lab), ptr(wrow), ptr(rep_hi), ptr(rep_lo), ptr(pm), ptr(pl),
         ptr(pO), ptr(lse), ptr(off), ptr(rowloss), None, ptr(drep), ptr(img), _st())
    call("ader_lbf_sum", ptr(rowloss), B, ptr(loss), _st())
    return loss, lse[:B].clone(), drep, rep_hi, rep_lo, off, lab, wrow, img


@logits_ce_x3.register_fake
def _(rep, emb, pos, ex_pos, N, w_train, w_ex):
    B = rep.shape[0]
    Bp = (B + 127) // 128 * 128
    bf = rep.new_empty(Bp * 168, dtype=torch.bfloat16)
    return (rep.new_empty(1), rep.new_empty(B), torch.empty_like(rep), bf, torch.empty_like(bf), rep.new_empty(Bp),
            pos.new_empty(Bp), rep.new_empty(Bp), rep.new_empty(Bp * 704, dtype=torch.uint8))


def _x3_setup(ctx, inputs, output):
    ctx.save_for_backward(output[2])
    ctx.n_in = len(inputs)


def _x3_backward(ctx, dloss, *unused):
    (drep,) = ctx.saved_tensors
    # d loss / d rep; the table's gradient is formed and consumed inside table_update_x3 (never materialised): None for emb
    return (drep * dloss,) + (None,) * (ctx.n_in - 1)


torch.library.register_autograd("ader::logits_ce_x3", _x3_backward, setup_context=_x3_setup)


@torch.library.custom_op("ader::logits_ce_x3_kd", mutates_args=())
def logits_ce_x3_kd(rep: torch.Tensor, emb: torch.Tensor, pos: torch.Tensor, ex_trow: torch.Tensor, teacher: torch.Tensor, N: int,
                    w_train: float, w_ex: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
                                                          torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """The adaptive-distillation loss of ADER.py:108-138 at float32 grade: rows [0, n_train) of rep are train rows (one-hot CE, weight
    w_train = 1/B_train), the rows after them exemplar rows distilled against softmax(teacher[ex_trow[e], :Np]) over the first Np items
    (weight w_ex = lambda/B_ex; the student logits are sliced before the softmax, ADER.py:134).  teacher [E_all, Np] fp32.  Returns loss,
    lse [B], drep [B,H], and for table_update_x3_kd: rep_hi, rep_lo, off, lab, wrow, trow, tlse2 in the padded row layout
    [train rows padded to 128 | exemplar rows padded to 128]."""
    _chk(rep, "rep", torch.float32, 2), _chk(emb, "emb", torch.float32, 2), _chk(pos, "pos", torch.int32, 1)
    _chk(ex_trow, "ex_trow", torch.int32, 1), _chk(teacher, "teacher", torch.float32, 2)
    B, H = rep.shape
    n_train, n_ex = pos.shape[0], ex_trow.shape[0]
    item_num, (E_all, Np) = emb.shape[0] - 1, teacher.shape
    Bt, Bk = (n_train + 127) // 128 * 128, (n_ex + 127) // 128 * 128
    Bp = Bt + Bk
    if (emb.shape[1] != H or H % 2 or H > 150 or not (1 <= Np <= N <= item_num) or n_train + n_ex != B or n_train < 1 or n_ex < 1
            or Bp > 4096):
        raise RuntimeError("ader::logits_ce_x3_kd: bad shapes (B=%d, n_train=%d, n_ex=%d, H=%d, Np=%d, N=%d, item_num=%d)"
                           % (B, n_train, n_ex, H, Np, N, item_num))
    _check_labels(pos, "logits_ce_x3_kd pos", N), _check_labels(ex_trow, "logits_ce_x3_kd ex_trow", E_all - 1)
    dev = rep.device
    i32 = dict(dtype=torch.int32, device=dev)
    tlse_all = torch.empty(E_all, device=dev)
    call("ader_row_lse", ptr(teacher), teacher.stride(0), Np, ptr(torch.arange(E_all, **i32)), E_all, ptr(tlse_all), _st())
    lab, trow = torch.empty(Bp, **i32), torch.empty(Bp, **i32)
    wrow, tlse2 = torch.empty(Bp, device=dev), torch.empty(Bp, device=dev)
    rep_hi, rep_lo, pm, pl, pO, lse, off, rowloss = _x3_fwd_scratch(N, Bp, dev)
    pO2 = torch.empty(call("ader_lx3_readout_ranges", Np, Bk) * Bk * 160, device=dev)
    loss, drep = torch.zeros(1, device=dev), torch.empty(B, H, device=dev)
    call("ader_lx3_fwd_kd", ptr(rep), ptr(emb), item_num, n_train, n_ex, Bt, Bp, H, N, Np, ptr(pos), ptr(ex_trow), ptr(teacher),
         teacher.stride(0), ptr(tlse_all), float(w_train), float(w_ex), ptr(lab), ptr(wrow), ptr(trow), ptr(tlse2), ptr(rep_hi),
         ptr(rep_lo), ptr(pm), ptr(pl), ptr(pO), ptr(pO2), ptr(lse), ptr(off), ptr(rowloss), ptr(loss), ptr(drep), _st())
    lse_c = torch.cat([lse[:n_train], lse[Bt:Bt + n_ex]])
    return loss, lse_c, drep, rep_hi, rep_lo, off, lab, wrow, trow, tlse2


@logits_ce_x3_kd.register_fake
def _(rep, emb, pos, ex_trow, teacher, N, w_train, w_ex):
    B = rep.shape[0]
    Bp = (pos.shape[0] + 127) // 128 * 128 + (ex_trow.shape[0] + 127) // 128 * 128
    bf = rep.new_empty(Bp * 168, dtype=torch.bfloat16)
    return (rep.new_empty(1), rep.new_empty(B), torch.empty_like(rep), bf, torch.empty_like(bf), rep.new_empty(Bp), pos.new_empty(Bp),
            rep.new_empty(Bp), pos.new_empty(Bp), rep.new_empty(Bp))


torch.library.register_autograd("ader::logits_ce_x3_kd", _x3_backward, setup_context=_x3_setup)


def _sparse_lists_x3(seq, lab, N):
    """The bucketed lists of the sparse table-gradient terms + the per-tile records of the x3 update (Engine._sparse_lists)."""
    dev = seq.device
    i32 = dict(dtype=torch.int32, device=dev)
    seq, lab = seq.reshape(-1).contiguous(), lab.reshape(-1).contiguous()
    n_sp, n_tg = seq.numel(), lab.numel()
    nb1 = call("ader_sparse_lists_starts", N)
    ids, order, tids, torder = torch.empty(n_sp, **i32), torch.empty(n_sp, **i32), torch.empty(n_tg, **i32), torch.empty(n_tg, **i32)
    sp_start, tg_start = torch.empty(nb1, **i32), torch.empty(nb1, **i32)
    scratch = torch.empty(call("ader_sparse_lists_scratch_n", n_sp, n_tg, N), **i32)
    call("ader_sparse_lists", ptr(seq), n_sp, ptr(lab), n_tg, N, ptr(scratch), ptr(ids), ptr(order), ptr(sp_start), ptr(tids), ptr(torder),
         ptr(tg_start), _st())
    meta = torch.empty(call("ader_tab_meta_ints", N), **i32)
    call("ader_tab_tile_meta", ptr(ids), ptr(order), ptr(sp_start), ptr(tids), ptr(torder), ptr(tg_start), N, ptr(meta), _st())
    return ids, order, tids, torder, meta


def _chk_table(emb, m, v, seq, g_rows, name):
    _chk(emb, "emb", torch.float32, 2), _chk(m, "m", torch.float32, 2), _chk(v, "v", torch.float32, 2)
    _chk(seq, "seq", torch.int32, 2), _chk(g_rows, "g_rows", torch.float32, 2)
    if m.shape != emb.shape or v.shape != emb.shape or g_rows.shape != (seq.numel(), emb.shape[1]):
        raise RuntimeError("ader::%s: m, v must match emb [V,H]; g_rows must be [B*T, H]" % name)


@torch.library.custom_op("ader::table_update_x3", mutates_args=("emb", "m", "v"))
def table_update_x3(emb: torch.Tensor, m: torch.Tensor, v: torch.Tensor, seq: torch.Tensor, g_rows: torch.Tensor, rep_hi: torch.Tensor,
                    rep_lo: torch.Tensor, off: torch.Tensor, lab: torch.Tensor, wrow: torch.Tensor, img: torch.Tensor, n_rows: int,
                    N: int, lr_t: float, beta1: float, beta2: float, eps: float) -> None:
    """Optimizer step of the item table (tf.train.AdamOptimizer applied densely, ADER.py:96) fused with the table gradient of
    ADER.py:91-93: for rows 1..N, dE = sum_b w_b (softmax_b - onehot_b) rep_b (recomputed on the matrix cores, never stored)
    + sqrt(H) x the input-embedding rows g_rows [B*T,H] (position p of seq contributes to row seq[p]; padding id 0 skipped), then
    TF-Adam in place on emb / m / v.  rep_hi .. img: as returned by logits_ce_x3 for the same batch of n_rows rows.
    lr_t = lr sqrt(1-b2^t)/(1-b1^t)."""
    _chk_table(emb, m, v, seq, g_rows, "table_update_x3")
    H = emb.shape[1]
    Bp = off.shape[0]
    if not (0 < n_rows <= Bp) or seq.shape[0] != n_rows:
        raise RuntimeError("ader::table_update_x3: n_rows must be the batch's row count (seq rows = %d, Bp = %d)" % (seq.shape[0], Bp))
    ids, order, tids, torder, meta = _sparse_lists_x3(seq, lab, N)
    call("ader_tab_update_x3", ptr(rep_hi), ptr(rep_lo), ptr(img), emb.shape[0] - 1, n_rows, Bp, H, N, ptr(off), ptr(ids), ptr(order),
         ids.numel(), ptr(g_rows), _sqrt_f32(H),
         ptr(tids), ptr(torder), tids.numel(), ptr(meta), ptr(wrow), ptr(emb), ptr(m), ptr(v), lr_t, beta1, beta2, eps, 0, -1, None, _st())


@torch.library.custom_op("ader::table_update_x3_kd", mutates_args=("emb", "m", "v"))
def table_update_x3_kd(emb: torch.Tensor, m: torch.Tensor, v: torch.Tensor, seq: torch.Tensor, g_rows: torch.Tensor,
                       rep_hi: torch.Tensor, rep_lo: torch.Tensor, off: torch.Tensor, lab: torch.Tensor, wrow: torch.Tensor,
                       teacher: torch.Tensor, trow: torch.Tensor, tlse2: torch.Tensor, n_train: int, N: int, lr_t: float, beta1: float,
                       beta2: float, eps: float) -> None:
    """table_update_x3 for a distilled step (ADER.py:132-137): the exemplar rows' dlogit is w (softmax(s[:Np]) - softmax(teacher row));
    arguments as logits_ce_x3_kd returned them."""
    _chk_table(emb, m, v, seq, g_rows, "table_update_x3_kd")
    _chk(teacher, "teacher", torch.float32, 2)
    H = emb.shape[1]
    Bp = off.shape[0]
    Bt = (n_train + 127) // 128 * 128
    img = torch.zeros(call("ader_x3_rep_image_bytes", Bp), dtype=torch.uint8, device=emb.device)
    call("ader_x3_rep_image", ptr(rep_hi), ptr(rep_lo), Bp, ptr(img), _st())
    ids, order, tids, torder, meta = _sparse_lists_x3(seq, lab, N)
    call("ader_tab_update_x3_kd", ptr(rep_hi), ptr(rep_lo), ptr(img), emb.shape[0] - 1, Bp, Bt, H, N, teacher.shape[1], ptr(off), ptr(ids),
         ptr(order), ids.numel(), ptr(g_rows), _sqrt_f32(H), ptr(tids), ptr(torder),
         tids.numel(), ptr(meta), ptr(wrow), ptr(teacher), teacher.stride(0), ptr(trow), ptr(tlse2), ptr(emb), ptr(m), ptr(v), lr_t, beta1,
         beta2, eps, _st())
