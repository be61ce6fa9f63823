"""CPU ORACLE (test infrastructure, NOT product code) for the ADER/SASRec hot path.

Op-for-op PyTorch-CPU restatement of the reference's TensorFlow graph:
  modules.py:23-50    normalize (LayerNorm, biased variance, eps 1e-8 inside the sqrt)
  modules.py:118-130  embedding (row 0 forced to zero, items scaled by sqrt(H))
  modules.py:163-223  multihead_attention (K/V from un-normalised x, key/causal/query masks,
                      -2**32+1 padding, softmax, dropout on probabilities, residual = LN'd queries)
  modules.py:252-266  feedforward (conv1d k=1 == dense; residual = LN'd input)
  ADER.py:25-93       forward, last-position representation, full-catalog logits, one-hot CE
  ADER.py:105-138     vanilla loss / ADER loss (CE on train rows + lambda * KD or one-hot on exemplars)
  ADER.py:99-103      rank prediction argsort(argsort(-logits))
  tf.train.AdamOptimizer (ADER.py:96) restated from TF 2.0's ApplyAdam kernel semantics.

PARITY STATUS: **unpinned** for the model math.  TensorFlow 2.0/2.1 (requirments.yaml:350-353) is
not installed here and the reference has no tests or golden vectors for this path,
so this file restates the published TF semantics at the reference's call sites; it is checked by
hand-derived cases and fp64 finite differences (tests/test_oracle_model.py), not by reference output.
What the reference itself publishes pins the path END TO END instead: the per-period test curves of its figure
(results.svg -> tests/golden/results_svg_curves.json) and its poster's ablation table, against which the HIP path that
matches this restatement op by op is run for all 16 periods of both datasets (tests/test_gpu_e2e_parity.py).
THIS FILE ITSELF is held to the same artefact (round 4; statistical pins, not bit-level ones): trained by itself on DIGINETICA
period 1 it lands on the figure's period-1 points, and run by itself through all 16 periods of the Finetune baseline it follows the
figure's Finetune curve; and ADER ITSELF -- this file's distillation loss + oracle/herding_ref.py, 16 periods, default flags -- reproduces
the figure's ADER curve period by period: averages 50.28 / 17.42 against 50.21 / 17.32, per-period mean deviation 0.20 / 0.12
(tests/golden/make_oracle_period1.py, make_oracle_finetune16.py, make_oracle_ader16.py; asserted by tests/test_oracle_model.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Dropout uses the build's counter-based mask spec (see dropout_keep) so the HIP path and the oracle
draw identical masks; TF's own RNG stream is not reproducible (SURVEY §7 "hard parts").
"""
import math

import numpy as np
import torch

NEG_PAD = float(-2 ** 32 + 1)  # modules.py:192,201 -> float32 -4294967296.0
LN_EPS = 1e-8                  # modules.py:24

SITE_EMB = 0


def site_attn(l):
    return 1 + 3 * l


def site_ffn1(l):
    return 2 + 3 * l


def site_ffn2(l):
    return 3 + 3 * l


# ----------------------------------------------------------------------------- dropout mask spec
def _lowbias32(x):
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def dropout_key(seed, step, site):
    with np.errstate(over="ignore"):
        a = _lowbias32(np.array([np.uint32(seed & 0xFFFFFFFF) ^ np.uint32(0x9E3779B9)], dtype=np.uint32))
        b = a + np.uint32(step & 0xFFFFFFFF) * np.uint32(0x85EBCA6B) + np.uint32(site) * np.uint32(0xC2B2AE35)
        return int(_lowbias32(b)[0])


def dropout_threshold(rate):
    """keep iff (hash >> 8) >= thr ; P(drop) = thr / 2**24."""
    return int(round(float(rate) * 16777216.0))


def dropout_keep(n_elem, first_index, seed, step, site, rate):
    """Boolean keep-mask for elements first_index .. first_index+n_elem-1 of a dropout site."""
    key = np.uint32(dropout_key(seed, step, site))
    idx = (np.arange(n_elem, dtype=np.uint64) + np.uint64(first_index)).astype(np.uint32)
    r = _lowbias32(idx ^ key)
    return (r >> np.uint32(8)) >= np.uint32(dropout_threshold(rate))


def _dropout(x, rate, training, seed, step, site, row0):
    """TF2 inverted dropout: x * (1/(1-rate)) * keep  (nn_ops.dropout_v2)."""
    if (not training) or rate == 0.0:
        return x
    per_row = int(np.prod(x.shape[1:]))
    keep = dropout_keep(x.numel(), row0 * per_row, seed, step, site, rate).reshape(tuple(x.shape))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    return (x * float(scale)) * torch.from_numpy(keep).to(x.dtype)


# ----------------------------------------------------------------------------- parameters
def param_shapes(item_num, T, H, L):
    shp = {"emb": (item_num + 1, H), "pos": (T, H)}
    for l in range(L):
        p = "b%d." % l
        shp.update({p + "ln1_g": (H,), p + "ln1_b": (H,),
                    p + "wq": (H, H), p + "bq": (H,), p + "wk": (H, H), p + "bk": (H,),
                    p + "wv": (H, H), p + "bv": (H,),
                    p + "ln2_g": (H,), p + "ln2_b": (H,),
                    p + "w1": (H, H), p + "b1": (H,), p + "w2": (H, H), p + "b2": (H,)})
    shp.update({"lnf_g": (H,), "lnf_b": (H,)})
    return shp


def init_params(item_num, T, H, L, seed=0, dtype=torch.float32):
    """TF defaults at the reference call sites: Glorot-uniform kernels/tables (tf.get_variable,
    tf.layers.dense, conv1d), zero biases, LN gamma=1 beta=0 (modules.py:45-46,119-123)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shp in param_shapes(item_num, T, H, L).items():
        base = name.split(".")[-1]
        if base in ("emb", "pos", "wq", "wk", "wv", "w1", "w2"):
            lim = math.sqrt(6.0 / (shp[0] + shp[1]))
            t = (torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * lim
        elif base.endswith("_g"):
            t = torch.ones(shp, dtype=torch.float64)
        else:
            t = torch.zeros(shp, dtype=torch.float64)
        out[name] = t.to(dtype)
    return out


# ----------------------------------------------------------------------------- forward
def layernorm(x, g, b):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)          # tf.nn.moments: biased
    return g * ((x - mean) / ((var + LN_EPS) ** 0.5)) + b    # modules.py:47-48


def forward_rep(params, seq, L, num_heads, *, training=False, rate=0.0, seed=0, step=0, row0=0,
                return_intermediates=False, relu_masks=None):
    """seq: int64/int32 [B,T].  Returns rep [B,H] (ADER.py:85).

    relu_masks (optional, test aid): {block: ("all", mask[B,T,H]) | ("last", mask[B,H])} replaces relu(a) by a*mask at
    the given rows.  ReLU is discontinuous in its derivative: an implementation whose pre-activations differ from this
    oracle's by rounding (e.g. the bf16x3 GEMMs, ~2^-16) takes the other branch for the handful of elements with
    |a| below that rounding, which moves gradients by O(1e-3) without being an error.  Feeding the implementation's own
    branch decisions makes the comparison exact again."""
    seq = torch.as_tensor(seq).long()
    B, T = seq.shape
    emb = params["emb"]
    dt = emb.dtype
    H = emb.shape[1]
    inter = {}
    mask = (seq != 0).to(dt).unsqueeze(-1)                                   # ADER.py:25
    table = torch.cat([torch.zeros(1, H, dtype=dt), emb[1:]], 0)             # modules.py:124-126
    x = table[seq] * float(np.float32(H ** 0.5)) if dt == torch.float32 else table[seq] * (H ** 0.5)
    x = x + params["pos"][:T].unsqueeze(0)                                   # ADER.py:41-52
    x = _dropout(x, rate, training, seed, step, SITE_EMB, row0)              # ADER.py:55-58
    x = x * mask                                                             # ADER.py:60
    inter["x0"] = x
    dh = H // num_heads
    for l in range(L):
        p = "b%d." % l
        q_in = layernorm(x, params[p + "ln1_g"], params[p + "ln1_b"])        # ADER.py:66
        Q = q_in @ params[p + "wq"] + params[p + "bq"]                       # modules.py:172
        K = x @ params[p + "wk"] + params[p + "bk"]                          # modules.py:173 (keys = raw x)
        V = x @ params[p + "wv"] + params[p + "bv"]
        Qh = Q.view(B, T, num_heads, dh).permute(0, 2, 1, 3)
        Kh = K.view(B, T, num_heads, dh).permute(0, 2, 1, 3)
        Vh = V.view(B, T, num_heads, dh).permute(0, 2, 1, 3)
        s = (Qh @ Kh.transpose(-1, -2)) / float(np.float32(dh ** 0.5))       # modules.py:182-185
        key_mask = torch.sign(torch.abs(x.sum(-1)))                          # modules.py:188 [B,T]
        s = torch.where(key_mask[:, None, None, :] == 0, torch.full_like(s, NEG_PAD), s)
        tril = torch.tril(torch.ones(T, T, dtype=dt))
        s = torch.where(tril[None, None] == 0, torch.full_like(s, NEG_PAD), s)   # modules.py:196-202
        a = torch.softmax(s, -1)                                             # modules.py:205
        query_mask = torch.sign(torch.abs(q_in.sum(-1)))                     # modules.py:208 (queries = LN(x))
        a = a * query_mask[:, None, :, None]
        a = _dropout(a, rate, training, seed, step, site_attn(l), row0)      # modules.py:214
        o = (a @ Vh).permute(0, 2, 1, 3).reshape(B, T, H)                    # modules.py:217-220
        x = o + q_in                                                         # modules.py:223
        inter["attn%d" % l] = x
        y = layernorm(x, params[p + "ln2_g"], params[p + "ln2_b"])           # ADER.py:77
        pre = y @ params[p + "w1"] + params[p + "b1"]
        h1 = torch.relu(pre)                                                 # modules.py:254-256
        if relu_masks is not None and l in relu_masks:
            kind, mk = relu_masks[l]
            mk = torch.as_tensor(mk).to(dt)
            if kind == "all":
                h1 = pre * mk.view(B, T, H)
            else:
                h1 = torch.cat([h1[:, :-1], (pre[:, -1] * mk.view(B, H)).unsqueeze(1)], 1)
        h1 = _dropout(h1, rate, training, seed, step, site_ffn1(l), row0)    # modules.py:257
        inter["h1d%d" % l] = h1
        h2 = h1 @ params[p + "w2"] + params[p + "b2"]                        # modules.py:259-261
        h2 = _dropout(h2, rate, training, seed, step, site_ffn2(l), row0)    # modules.py:262
        x = (h2 + y) * mask                                                  # modules.py:266, ADER.py:80
        inter["blk%d" % l] = x
    x = layernorm(x, params["lnf_g"], params["lnf_b"])                       # ADER.py:82
    rep = x[:, -1, :]                                                        # ADER.py:85
    if return_intermediates:
        inter["final"] = x
        return rep, inter
    return rep


def _bf16_ste(x):
    """Round to bfloat16 with a straight-through gradient (models the bf16-MFMA logits mode of the HIP path)."""
    return x + (x.to(torch.bfloat16).to(x.dtype) - x).detach()


def logits_from_rep(params, rep, max_item, logits_bf16=False):
    item_emb = params["emb"][1:max_item + 1]                                 # ADER.py:91 (unscaled table)
    if logits_bf16:
        return _bf16_ste(rep) @ _bf16_ste(item_emb).t()
    return rep @ item_emb.t()                                                # ADER.py:92


def loss_fn(params, seq, pos, max_item, L, num_heads, *, ex_logits=None, ex_pos=None, lambda_=0.0,
            training=True, rate=0.0, seed=0, step=0, row0=0, n_train_global=None, n_ex_global=None, logits_bf16=False,
            relu_masks=None):
    """Vanilla loss (ADER.py:93) or ADER loss (ADER.py:108-137).

    seq holds the train rows first, exemplar rows after (main.py:229); the split point is inferred
    from the exemplar feed (ADER.py:113-115).  n_*_global override the mean denominators for the
    data-parallel shards (each rank scales its local sums by the global counts)."""
    rep = forward_rep(params, seq, L, num_heads, training=training, rate=rate, seed=seed, step=step, row0=row0,
                      relu_masks=relu_masks)
    logits = logits_from_rep(params, rep, max_item, logits_bf16)
    n_ex = 0 if (ex_logits is None and ex_pos is None) else (len(ex_logits) if ex_logits is not None else len(ex_pos))
    n_train = seq.shape[0] - n_ex
    pos = torch.as_tensor(pos).long()
    lsm = torch.log_softmax(logits[:n_train], -1)
    ce = -lsm[torch.arange(n_train), pos - 1]                                # one_hot(pos-1)
    loss = ce.sum() / float(n_train_global if n_train_global is not None else n_train)
    if n_ex:
        den = float(n_ex_global if n_ex_global is not None else n_ex)
        if ex_logits is not None:                                            # ADER.py:132-137
            tl = torch.as_tensor(ex_logits).to(logits.dtype)
            Np = tl.shape[1]
            student = logits[n_train:, :Np]                                  # sliced BEFORE the softmax
            teacher = torch.softmax(tl, -1)
            kd = -(teacher * torch.log_softmax(student, -1)).sum(-1)
            loss = loss + lambda_ * (kd.sum() / den)
        else:                                                                # ADER.py:126-131
            ep = torch.as_tensor(ex_pos).long()
            lsm_e = torch.log_softmax(logits[n_train:], -1)
            loss = loss + lambda_ * ((-lsm_e[torch.arange(n_ex), ep - 1]).sum() / den)
    return loss


def loss_and_grads(params, *args, **kw):
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    loss = loss_fn(ps, *args, **kw)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in ps.items()}
    return loss.detach(), grads


# ----------------------------------------------------------------------------- Adam (TF ApplyAdam)
class TFAdam:
    """tf.train.AdamOptimizer(lr) defaults beta1=.9 beta2=.999 eps=1e-8 (ADER.py:96):
       lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= lr_t*m/(sqrt(v)+eps).
       beta powers are float32 running products, as TF keeps them in float32 variables."""

    def __init__(self, params, beta1=0.9, beta2=0.999, eps=1e-8):
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.b1p = np.float32(beta1)
        self.b2p = np.float32(beta2)
        self.t = 0

    def lr_t(self, lr):
        return float(np.float32(lr) * np.sqrt(np.float32(1) - self.b2p) / (np.float32(1) - self.b1p))

    def step(self, params, grads, lr):
        a = self.lr_t(lr)
        omb1 = float(np.float32(1) - np.float32(self.b1))
        omb2 = float(np.float32(1) - np.float32(self.b2))
        with torch.no_grad():
            for k, p in params.items():
                g = grads[k]
                self.m[k] += (g - self.m[k]) * omb1
                self.v[k] += (g * g - self.v[k]) * omb2
                p -= (self.m[k] * a) / (self.v[k].sqrt() + self.eps)
        self.b1p = np.float32(self.b1p * np.float32(self.b1))
        self.b2p = np.float32(self.b2p * np.float32(self.b2))
        self.t += 1


def train_step(params, opt, seq, pos, max_item, L, num_heads, lr, **kw):
    loss, grads = loss_and_grads(params, seq, pos, max_item, L, num_heads, **kw)
    opt.step(params, grads, lr)
    return float(loss)


# ----------------------------------------------------------------------------- ranking / metrics
def rank_all(params, seq, max_item, L, num_heads):
    """pred_last = argsort(argsort(-logits)) (ADER.py:103): 0-based rank of every item; ties -> lower index first."""
    with torch.no_grad():
        rep = forward_rep(params, seq, L, num_heads)
        logits = logits_from_rep(params, rep, max_item)
        order = torch.argsort(-logits, dim=-1, stable=True)
        return torch.argsort(order, dim=-1, stable=True)


def rank_of_target(logits, target):
    """Rank of item `target` (1-based id) in one logits row = #greater + #equal with lower index."""
    logits = np.asarray(logits)
    t = logits[target - 1]
    return int((logits > t).sum() + (logits[: target - 1] == t).sum())


def metrics(ranks):
    """Evaluator.results (util.py:329-339): (MRR@20, RECALL@20, MRR@10, RECALL@10)."""
    n = len(ranks)
    r20 = [r for r in ranks if r < 20]
    r10 = [r for r in ranks if r < 10]
    return (sum(1.0 / (r + 1) for r in r20) / n, len(r20) / n,
            sum(1.0 / (r + 1) for r in r10) / n, len(r10) / n)
