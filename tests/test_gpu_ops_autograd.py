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


def test_one_step_through_torch_ops_matches_the_engine():
    import ader_amd.ops  # noqa: F401  (registers torch.ops.ader.*)
    from ader_amd.engine import Engine
    eng = Engine(ITEMS, maxlen=T, hidden_units=H, num_blocks=L, num_heads=HEADS, seed=5, logits_dtype="f32", gemm="f32")
    g = torch.Generator().manual_seed(1)
    for k in eng.layout:                 # LN beta away from 0: the query mask sign(|sum LN(x)|) must not hinge on rounding noise
        if k.endswith("_b"):
            eng.param(k).copy_(torch.randn(eng.layout[k][1], generator=g) * 0.1)
    seq_h, pos_h = _batch()
    loss_e = float(eng.loss_and_grad(seq_h, pos_h, N, rate=0.0).item())
    torch.cuda.synchronize()

    p = {k: eng.param(k).detach().clone().requires_grad_(True) for k in eng.layout}
    seq = torch.from_numpy(seq_h).cuda()
    lab = torch.from_numpy(pos_h).cuda()
    A = torch.ops.ader
    x = A.embed_fwd(seq, p["emb"], p["pos"], 0, 0, 1.0).reshape(B * T, H)
    for l in range(L):
        b = "b%d." % l
        q_in, _, _, kmask, qmask = A.layernorm(x, p[b + "ln1_g"], p[b + "ln1_b"])
        x1 = A.attn_fwd(x, q_in, p[b + "wq"], p[b + "bq"], p[b + "wk"], p[b + "bk"], p[b + "wv"], p[b + "bv"], kmask, qmask, B, T,
                        HEADS, 0, 0, 1.0)[0]
        y = A.layernorm(x1, p[b + "ln2_g"], p[b + "ln2_b"])[0]
        x = A.ffn_fwd(y, p[b + "w1"], p[b + "b1"], p[b + "w2"], p[b + "b2"], seq, 0, 0, 1.0, 0, 0, 1.0)[0]
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
