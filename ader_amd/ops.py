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
"""
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
    D = torch.empty(max(n * H, 1), device=dev)
    chosen = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    sel = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(max(G, 1), dtype=torch.int32, device=dev)
    call("ader_herding_select", ptr(rep), ptr(seg), ptr(quota), ptr(max_steps), G, n, H, ptr(D), ptr(chosen), ptr(sel), ptr(cnt), None,
         _st())
    return sel[:n], cnt[:G]


@herding_select.register_fake
def _(rep, seg, quota, max_steps):
    return quota.new_empty(rep.shape[0]), quota.new_empty(quota.shape[0])
