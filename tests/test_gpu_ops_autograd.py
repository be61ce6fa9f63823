"""torch.ops.ader.* as a trainable graph (SURVEY 8b "Native ABI"): one SASRec step composed ONLY of the registered operators
(embed_fwd, layernorm, attn_fwd, ffn_fwd, logits_ce) plus tensor views, differentiated by torch.autograd through their registered
backward operators, must reproduce the loss and every parameter gradient of Engine.loss_and_grad (the ctypes product path) on the
same exact-f32 kernels -- and one ader::adam_step on top must equal Engine.adam."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ITEMS, T, H, L, HEADS, B, N = 700, 50, 150, 2, 1, 48, 650


def _batch():
    rs = np.random.RandomState(3)
    seq = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        ln = rs.randint(1, T + 1)
        seq[b, T - ln:] = rs.randint(1, N + 1, size=ln)
    return seq, rs.randint(1, N + 1, size=B).astype(np.int32)


@pytest.mark.parametrize("rate", [0.0, 0.3])
def test_one_step_through_torch_ops_matches_the_engine(rate):
    """rate 0.3: every dropout site (prologue, attention probabilities, both FFN sites) is ON, with the engine's counter keys, so
    the backward operators' mask REPLAY (embed_bwd, attn_bwd, ffn_bwd recompute the masks instead of storing them) is checked too."""
    import ader_amd.ops  # noqa: F401  (registers torch.ops.ader.*)
    from ader_amd.engine import Engine, SITE_EMB, dropout_key, site_attn, site_ffn1, site_ffn2
    eng = Engine(ITEMS, maxlen=T, hidden_units=H, num_blocks=L, num_heads=HEADS, seed=5, logits_dtype="f32", gemm="f32")
    g = torch.Generator().manual_seed(1)
    for k in eng.layout:                 # LN beta away from 0: the query mask sign(|sum LN(x)|) must not hinge on rounding noise
        if k.endswith("_b"):
            eng.param(k).copy_(torch.randn(eng.layout[k][1], generator=g) * 0.1)
    seq_h, pos_h = _batch()
    eng.global_step = 3
    loss_e = float(eng.loss_and_grad(seq_h, pos_h, N, rate=rate).item())
    torch.cuda.synchronize()
    thr = int(round(rate * 16777216.0))
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate))) if rate else 1.0

    def dk(site):                        # (key, threshold, scale) of a site as Engine._drop builds them (seed 5, step 3)
        return (dropout_key(5, 3, site), thr, scale) if rate else (0, 0, 1.0)

    p = {k: eng.param(k).detach().clone().requires_grad_(True) for k in eng.layout}
    seq = torch.from_numpy(seq_h).cuda()
    lab = torch.from_numpy(pos_h).cuda()
    A = torch.ops.ader
    x = A.embed_fwd(seq, p["emb"], p["pos"], *dk(SITE_EMB)).reshape(B * T, H)
    for l in range(L):
        b = "b%d." % l
        q_in, _, _, kmask, qmask = A.layernorm(x, p[b + "ln1_g"], p[b + "ln1_b"])
        x1 = A.attn_fwd(x, q_in, p[b + "wq"], p[b + "bq"], p[b + "wk"], p[b + "bk"], p[b + "wv"], p[b + "bv"], kmask, qmask, B, T,
                        HEADS, *dk(site_attn(l)))[0]
        y = A.layernorm(x1, p[b + "ln2_g"], p[b + "ln2_b"])[0]
        x = A.ffn_fwd(y, p[b + "w1"], p[b + "b1"], p[b + "w2"], p[b + "b2"], seq, *dk(site_ffn1(l)), *dk(site_ffn2(l)))[0]
    x_last = x.view(B, T, H)[:, T - 1, :].contiguous()
    rep = A.layernorm(x_last, p["lnf_g"], p["lnf_b"])[0]
    w = torch.full((B,), 1.0 / B, device="cuda")
    loss = A.logits_ce(rep, p["emb"], lab, w, N)[0]
    loss.sum().backward()
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - loss_e) < 1e-5 * max(1.0, abs(loss_e)), (float(loss.item()), loss_e)
    for k in eng.layout:
        ge = eng.gradient(k).cpu().numpy()
        go = p[k].grad.cpu().numpy()
        if k == "emb":
            ge, go = ge[:N + 1], go[:N + 1]
        # (floor: the key bias has a mathematically zero gradient -- a constant added to every score of a query -- and both paths
        #  return rounding noise for it, as in tests/test_gpu_parity.py)
        err = np.abs(ge - go).max() / max(np.abs(ge).max(), 1e-4)
        assert err < 2e-4, (k, err)

    # one optimiser step through ader::adam_step == Engine.adam
    flat = torch.cat([p[k].detach().reshape(-1) for k in ("pos",)])
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    gflat = p["pos"].grad.reshape(-1).contiguous()
    lr_t = eng._lr_t(5e-4)
    A.adam_step(flat, m, v, gflat, lr_t, 0.9, 0.999, 1e-8)
    eng.adam(5e-4)
    torch.cuda.synchronize()
    assert torch.allclose(flat, eng.param("pos").reshape(-1), atol=2e-7)


def test_ops_reject_bad_inputs():
    import ader_amd.ops  # noqa: F401
    x = torch.zeros(4, 150, device="cuda")
    with pytest.raises(RuntimeError):
        torch.ops.ader.layernorm(x.double(), torch.ones(150, device="cuda"), torch.zeros(150, device="cuda"))
    with pytest.raises(RuntimeError):
        torch.ops.ader.layernorm(x, torch.ones(149, device="cuda"), torch.zeros(150, device="cuda"))
    with pytest.raises(RuntimeError):
        torch.ops.ader.logits_ce(x, torch.zeros(10, 150, device="cuda"), torch.ones(4, dtype=torch.int32, device="cuda"),
                                 torch.ones(4, device="cuda"), 50)


# ------------------------------------------------------------------------------------------------ the fast path as operators
def _x3_engine(seed=5):
    from ader_amd.engine import Engine
    eng = Engine(ITEMS, maxlen=T, hidden_units=H, num_blocks=L, num_heads=HEADS, seed=seed, logits_dtype="x3")
    g = torch.Generator().manual_seed(3)
    for k in eng.layout:
        if k.endswith("_b"):
            eng.param(k).copy_(torch.randn(eng.layout[k][1], generator=g) * 0.1)
    eng.refresh_shadow()
    return eng


@pytest.mark.parametrize("mode", ["vanilla", "onehot_ex", "kd"])
def test_x3_fast_path_ops_step_the_table_like_the_engine_bitwise(mode):
    """torch.ops.ader.logits_ce_x3[_kd] (flash forward at float32 grade: loss, lse, dRep) + torch.ops.ader.table_update_x3[_kd] (table
    gradient + dense TF-Adam fused) against Engine.train_step on the same batch: the loss and the item table's theta / Adam m / Adam v
    after the step are bit-identical (the ops launch the same kernels with the same operands); autograd through logits_ce_x3 hands
    back dRep.  Reference: ADER.py:88-96 (vanilla), :126-131 (one-hot exemplars), :132-137 (distillation)."""
    import ader_amd.ops  # noqa: F401
    rs = np.random.RandomState(31)
    B, N, Np, n_ex = 150, 600, 520, 37
    n_train = B - (0 if mode == "vanilla" else n_ex)
    seq = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        ln = int(rs.randint(1, T + 1))
        seq[b, T - ln:] = rs.randint(1, (Np if b >= n_train else N) + 1, size=ln)
    seq[3, -2:] = seq[4, -1]
    pos = rs.randint(1, N + 1, size=n_train).astype(np.int32)
    pos[0], pos[1] = N, pos[2]
    lam, lr, rate = 0.7, 5e-4, 0.3
    kw = {}
    ex_pos = np.zeros(0, dtype=np.int32)
    teacher = trow = None
    if mode == "onehot_ex":
        ex_pos = rs.randint(1, N + 1, size=n_ex).astype(np.int32)
        kw = dict(ex_pos=ex_pos, lambda_=lam)
    elif mode == "kd":
        teacher = torch.from_numpy(rs.standard_normal((50, Np)).astype(np.float32)).cuda()
        trow = rs.randint(0, 50, size=n_ex).astype(np.int32)
        kw = dict(teacher=teacher, ex_trow=trow, lambda_=lam)
    a = _x3_engine()
    a.global_step = 4
    loss_a = a.train_step(seq, pos, N, lr, rate=rate, **kw)
    torch.cuda.synchronize()

    b = _x3_engine()
    b.global_step = 4
    b.late_side_stream = False                           # (small reductions issued in line: nothing queued behind the ops)
    dev = b.device
    seq_d = torch.from_numpy(seq).to(dev)
    pos_d = torch.from_numpy(pos).to(dev)
    b._refresh_stream()
    rep = b.forward(seq_d, training=True, rate=rate, step=b.global_step, save=True).detach().clone().requires_grad_(True)
    emb, m, v = b.param("emb"), b.view(b.adam_m, "emb"), b.view(b.adam_v, "emb")
    w_train, w_ex = 1.0 / n_train, (lam / n_ex if mode != "vanilla" else 0.0)
    if mode == "kd":
        out = torch.ops.ader.logits_ce_x3_kd(rep, emb, pos_d, torch.from_numpy(trow).to(dev), teacher, N, w_train, w_ex)
        loss_b, lse, drep, rep_hi, rep_lo, off, lab, wrow, trw, tlse2 = out
    else:
        out = torch.ops.ader.logits_ce_x3(rep, emb, pos_d, torch.from_numpy(ex_pos).to(dev), N, w_train, w_ex)
        loss_b, lse, drep, rep_hi, rep_lo, off, lab, wrow, img = out
    assert float(loss_b.item()) == float(loss_a.item())
    (g_rep,) = torch.autograd.grad(loss_b.sum(), rep)                      # autograd formula: d loss / d rep = dRep
    assert torch.equal(g_rep, drep)
    dx = b._blocks_backward(seq_d, drep.detach(), True, None)              # per-position input-gradient rows stay in dx
    lr_t = b._lr_t(lr)
    if mode == "kd":
        torch.ops.ader.table_update_x3_kd(emb, m, v, seq_d, dx, rep_hi, rep_lo, off, lab, wrow, teacher, trw, tlse2, n_train, N, lr_t,
                                          b.beta1, b.beta2, b.eps)
    else:
        torch.ops.ader.table_update_x3(emb, m, v, seq_d, dx, rep_hi, rep_lo, off, lab, wrow, img, B, N, lr_t, b.beta1, b.beta2, b.eps)
    torch.cuda.synchronize()
    for name, buf in (("theta", "theta"), ("m", "adam_m"), ("v", "adam_v")):
        ta, tb = a.view(getattr(a, buf), "emb"), b.view(getattr(b, buf), "emb")
        assert torch.equal(ta, tb), name
    assert float(a.view(a.adam_v, "emb")[1:N + 1].abs().max()) > 0
    # a label outside [0, N] never degrades the loss silently to lse-only: by default (ops.STRICT_LABELS) the op that receives it raises;
    # with strict off it is flagged on the device (no host synchronisation inside the op: stream-ordered, capturable) and raises at the
    # next ops.check_status(), which clears the flag.  0 = "no target" (a weight-0 padding row) is legal
    ader_amd.ops.check_status()
    bad = pos_d.clone()
    bad[0] = N + 1

    def call_bad():
        if mode != "kd":
            return torch.ops.ader.logits_ce_x3(rep.detach(), emb, bad, torch.from_numpy(ex_pos).to(dev), N, w_train, w_ex)
        return torch.ops.ader.logits_ce_x3_kd(rep.detach(), emb, bad, torch.from_numpy(trow).to(dev), teacher, N, 1.0, 1.0)
    assert ader_amd.ops.STRICT_LABELS
    with pytest.raises(RuntimeError):
        call_bad()
    ader_amd.ops.STRICT_LABELS = False
    try:
        call_bad()
        with pytest.raises(RuntimeError):
            ader_amd.ops.check_status()
        ader_amd.ops.check_status()                                        # (the flag was cleared by the raise)
    finally:
        ader_amd.ops.STRICT_LABELS = True
