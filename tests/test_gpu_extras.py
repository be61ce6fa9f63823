"""GPU tests of the pieces around the hot path that SURVEY 8(f)#4 lists: the `loss` exemplar selector with a true per-row loss
(reference util.py:463-495), a mid-period resume through Saver + ExemplarStore (main.py:209-213, 280, 283, 312), and the
torch.library view of the launchers (torch.ops.ader.*)."""
import os
import sys
import tempfile
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ader_ref_cpu as R  # noqa: E402


def _args(**kw):
    a = types.SimpleNamespace(maxlen=50, hidden_units=150, num_blocks=2, num_heads=1, random_seed=0, dropout_rate=0.3, l2_emb=0.0,
                              disable_distillation=False, logits_dtype="bf16")
    a.__dict__.update(kw)
    return a


def _seqs(rs, B, T, n_items):
    seq = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        ln = int(rs.randint(1, T + 1))
        seq[b, T - ln:] = rs.randint(1, n_items + 1, size=ln)
    return seq


def test_row_losses_match_oracle_and_loss_selection_ranks_by_them():
    """Engine.row_losses == the oracle's per-row -log softmax(logits)[label] (eval mode, float32 kernels: 2e-5), and
    ExemplarGenerator.loss_selection keeps, per label, the min(quota, n) candidates of smallest loss in stable order."""
    from ader_amd.model import Ader
    from ader_amd.exemplar import ExemplarGenerator
    item_num, N, T = 400, 350, 50
    model = Ader(item_num, _args(logits_dtype="f32"))
    eng = model.engine
    g = torch.Generator().manual_seed(3)
    for k in eng.layout:
        if k.endswith("_b"):
            eng.param(k).copy_(torch.randn(eng.layout[k][1], generator=g) * 0.1)
    rs = np.random.RandomState(0)
    n = 300
    seq = _seqs(rs, n, T, N)
    pos = rs.randint(1, 40, size=n).astype(np.int32)            # few labels: groups with several candidates
    got = eng.row_losses(seq, pos, N).cpu().numpy()
    p64 = {k: v.double() for k, v in eng.export_params().items()}
    with torch.no_grad():
        rep = R.forward_rep(p64, seq, 2, 1)
        lsm = torch.log_softmax(R.logits_from_rep(p64, rep, N), -1)
        want = -lsm[torch.arange(n), torch.as_tensor(pos).long() - 1].numpy()
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    data = [list(s[s != 0]) + [int(p)] for s, p in zip(seq, pos)]
    np.random.seed(1)
    gen = ExemplarGenerator(data, 60, False, 256, T, 0.3, N)
    saved = gen.loss_selection(None, model)
    assert saved == len(gen.store) and saved > 0
    # every label: the stored sessions are its candidates of smallest oracle loss (ties / 2e-5-close losses may swap neighbours)
    by_label = {}
    for i, p in enumerate(pos):
        by_label.setdefault(int(p), []).append(i)
    for label, ex in gen.exemplars.items():
        cand = by_label[label]
        m = min(int(gen.item_count[label - 1]), len(cand))
        assert len(ex) == m
        kept = sorted(want[i] for i in cand)[:m]
        stored = []
        for sess, _ in ex:
            match = [i for i in cand if list(seq[i][seq[i] != 0]) + [label] == sess]
            stored.append(min(want[i] for i in match))
        assert np.allclose(sorted(stored), kept, atol=1e-4)
    for label in by_label:
        if int(gen.item_count[label - 1]) == 0:
            assert label not in gen.exemplars or len(gen.exemplars[label]) == 0     # `if m < 0.5: continue` (util.py:481)


def test_mid_period_resume_is_bitwise():
    """Stop in the middle of a distilled period, write the checkpoint (Saver) and the exemplar store (ExemplarStore.save), restore
    both into a fresh model and continue: parameters, Adam state and loss equal the uninterrupted run bit for bit (the reference
    keeps exemplars in memory only and cannot resume, main.py:312)."""
    from ader_amd.model import Ader, Saver, Session
    from ader_amd.exemplar import ExemplarStore
    item_num, N, Np, T, B, n_ex = 3000, 2800, 2500, 50, 200, 40
    rs = np.random.RandomState(7)
    ex_rows = np.concatenate([_seqs(rs, 64, T, Np), rs.randint(1, Np + 1, size=(64, 1)).astype(np.int32)], 1)
    batches = [(_seqs(rs, B + n_ex, T, N), rs.randint(1, N + 1, size=B).astype(np.int32), rs.randint(0, 64, size=n_ex).astype(np.int32))
               for _ in range(6)]

    def fresh():
        m = Ader(item_num, _args())
        m.update_loss(lambda_=0.6)
        return m

    def steps(m, store, lo, hi):
        for seq, pos, trow in batches[lo:hi]:
            m.train_step(seq, pos, N, 5e-4, 0.3, teacher=store.logits, ex_trow=trow)
        torch.cuda.synchronize()

    a = fresh()
    store = ExemplarStore(ex_rows, a.engine.teacher_logits(ex_rows[:, :T], Np), Np)
    steps(a, store, 0, 6)
    want = (a.engine.theta.clone(), a.engine.adam_m.clone(), a.engine.adam_v.clone(), float(a.engine.loss.item()))

    b = fresh()
    store_b = ExemplarStore(ex_rows, b.engine.teacher_logits(ex_rows[:, :T], Np), Np)
    steps(b, store_b, 0, 3)
    with tempfile.TemporaryDirectory() as d:
        Saver(b).save(Session(b), os.path.join(d, "mid.ckpt"))
        store_b.save(os.path.join(d, "exemplars.pt"))
        del b, store_b
        c = fresh()
        c.engine.init_params(123)                                # a different state, to be overwritten by the restore
        Saver(c).restore(Session(c), os.path.join(d, "mid.ckpt"))
        store_c = ExemplarStore.load(os.path.join(d, "exemplars.pt"), device=c.engine.device)
    assert np.array_equal(store_c.rows, ex_rows) and store_c.max_item == Np and c.engine.global_step == 3
    steps(c, store_c, 3, 6)
    got = (c.engine.theta, c.engine.adam_m, c.engine.adam_v, float(c.engine.loss.item()))
    for x, y in zip(want[:3], got[:3]):
        assert torch.equal(x, y)
    assert want[3] == got[3]


def test_torch_ops_match_the_engine():
    """torch.ops.ader.* (ader_amd/ops.py) are the same launchers behind torch's operator registry: results equal the engine's."""
    import ader_amd.ops  # noqa: F401  (registers the operators)
    from ader_amd.engine import Engine
    item_num, N, T, H, B = 500, 450, 50, 150, 48
    eng = Engine(item_num, maxlen=T, hidden_units=H, num_blocks=2, num_heads=1, seed=0, logits_dtype="bf16")
    rs = np.random.RandomState(1)
    seq = _seqs(rs, B, T, N)
    pos = rs.randint(1, N + 1, size=B).astype(np.int32)
    dseq, dpos = torch.from_numpy(seq).cuda(), torch.from_numpy(pos).cuda()
    rep = eng.encode(seq)
    # rank_of_target == Engine.rank_targets
    rk = torch.ops.ader.rank_of_target(rep, eng.param("emb").contiguous(), dpos, N)
    assert np.array_equal(rk.cpu().numpy(), eng.rank_targets(seq, pos, N))
    # embed_fwd without dropout == the table rows * sqrt(H) + positions, masked
    x = torch.ops.ader.embed_fwd(dseq, eng.param("emb").contiguous(), eng.param("pos").contiguous(), 0, 0, 1.0)
    ref = (eng.param("emb")[dseq.long()] * float(np.sqrt(np.float32(H))) + eng.param("pos")[None]) * (dseq != 0)[..., None]
    assert torch.allclose(x, ref, atol=1e-6)
    # layernorm_fwd
    y, mean, std = torch.ops.ader.layernorm_fwd(x.view(-1, H).contiguous(), eng.param("lnf_g").contiguous(), eng.param("lnf_b").contiguous())
    xr = x.view(-1, H).double()
    mu, var = xr.mean(-1, keepdim=True), xr.var(-1, unbiased=False, keepdim=True)
    yr = (xr - mu) / (var + 1e-8).sqrt() * eng.param("lnf_g").double() + eng.param("lnf_b").double()
    real = (dseq.view(-1) != 0)
    assert torch.allclose(y[real].double(), yr[real], atol=1e-4)
    # logits_ce_fwd: mean CE == the oracle with bf16 operand rounding
    w = torch.full((B,), 1.0 / B, device="cuda")
    loss, lse, drep, off = torch.ops.ader.logits_ce_fwd(rep.contiguous(), eng.shadow, dpos, w, N)
    p64 = {k: v.double() for k, v in eng.export_params().items()}
    lg = R.logits_from_rep(p64, rep.cpu().double(), N, logits_bf16=True)
    want = float(-torch.log_softmax(lg, -1)[torch.arange(B), torch.as_tensor(pos).long() - 1].mean())
    assert abs(float(loss.item()) - want) < 3e-4 * max(1.0, abs(want))
    # adam_step == TF ApplyAdam
    g = torch.Generator().manual_seed(0)
    p, gr = torch.randn(1000, generator=g).cuda(), torch.randn(1000, generator=g).cuda() * 0.01
    m, v = torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    p0 = p.clone()
    torch.ops.ader.adam_step(p, m, v, gr, 1e-3, 0.9, 0.999, 1e-8)
    mr, vr = 0.1 * gr.double(), 0.001 * gr.double() ** 2
    assert torch.allclose(p.double(), p0.double() - 1e-3 * mr / (vr.sqrt() + 1e-8), atol=1e-6)
    # dtype / shape violations raise instead of clamping
    with pytest.raises(RuntimeError):
        torch.ops.ader.rank_of_target(rep, eng.param("emb").contiguous(), dpos.long(), N)


def test_ewc_fisher_and_penalty_match_oracle():
    """EWC baseline (reference EWC.py:115-164).  (a) compute_fisher: mean of squared per-sample gradients of the eval-mode cross
    entropy, against the oracle's autograd, every parameter tensor within 2e-3 normalised (twice the gradient tolerance: squares).
    (b) one training step with the penalty lambda/2 sum F (theta - prev)^2: loss and the whole gradient (read back from Adam's
    first-moment state, m = 0.1 g after one step) against oracle gradient + lambda F (theta - prev)."""
    from ader_amd.engine import Engine
    item_num, N, T, H, L = 300, 260, 50, 150, 2
    eng = Engine(item_num, maxlen=T, hidden_units=H, num_blocks=L, num_heads=1, seed=2, logits_dtype="f32", gemm="f32")
    g = torch.Generator().manual_seed(4)
    for k in eng.layout:
        if k.endswith("_b"):
            eng.param(k).copy_(torch.randn(eng.layout[k][1], generator=g) * 0.1)
    rs = np.random.RandomState(3)
    n = 12
    seq, pos = _seqs(rs, n, T, N), rs.randint(1, N + 1, size=n).astype(np.int32)
    p64 = {k: v.double() for k, v in eng.export_params().items()}
    F = eng.compute_fisher(seq, pos, N)
    torch.cuda.synchronize()
    want = {k: torch.zeros_like(v) for k, v in p64.items()}
    for i in range(n):
        _, og = R.loss_and_grads(p64, seq[i:i + 1], pos[i:i + 1], N, L, 1, training=False)
        for k in want:
            want[k] += og[k] ** 2 / n
    for k in eng.layout:
        got = eng.view(F, k).cpu().double()
        err = float((got - want[k]).abs().max() / max(float(want[k].abs().max()), 1e-12))
        assert err < 2e-3, (k, err)
    # (b) penalty step
    lam = 3.0
    Fr = torch.zeros(eng.P, device="cuda")            # (the flat buffer has alignment gaps between the tensors: F stays 0 there,
    prev = eng.theta.clone()                           #  as it does when it comes from compute_fisher)
    for k in eng.layout:
        eng.view(Fr, k).copy_(torch.rand(eng.layout[k][1], generator=g) * 5.0)
        eng.view(prev, k).add_((torch.randn(eng.layout[k][1], generator=g) * 0.01).cuda())
    eng.ewc = {"F": Fr, "prev": prev, "lam": lam}
    B = 40
    seq2, pos2 = _seqs(rs, B, T, N), rs.randint(1, N + 1, size=B).astype(np.int32)
    loss = eng.train_step(seq2, pos2, N, 5e-4, rate=0.0)
    torch.cuda.synchronize()
    ol, og = R.loss_and_grads(p64, seq2, pos2, N, L, 1, training=True, rate=0.0, seed=2, step=0)
    pen = 0.0
    for k in eng.layout:
        d = p64[k] - eng.view(prev, k).cpu().double()
        f = eng.view(Fr, k).cpu().double()
        pen += float((f * d * d).sum())
        wantg = og[k] + lam * f * d
        gotg = eng.view(eng.adam_m, k).cpu().double() / 0.1
        err = float((gotg - wantg).abs().max() / max(float(wantg.abs().max()), 1e-12))
        assert err < 5e-4, (k, err)
    assert abs(float(loss.item()) - (float(ol) + 0.5 * lam * pen)) < 1e-4 * max(1.0, abs(float(ol) + 0.5 * lam * pen))


def test_ewc_driver_two_periods():
    """python -m ader_amd.main --ewc=True (reference main.py --ewc): period 1 vanilla, exemplars selected, Fisher information computed;
    period 2 trains with the penalty.  Plumbing: flags, flow, sane metrics."""
    from ader_amd import main as M
    with tempfile.TemporaryDirectory() as d:
        args = M.build_parser().parse_args(["--dataset", "DIGINETICA", "--ewc", "True", "--lambda_", "100", "--max_periods", "2",
                                            "--num_epochs", "1", "--ewc_sample_num", "64", "--results_root", d])
        lines = []
        out = M.run(args, log=lambda s="": lines.append(str(s)))
    assert args.dropout_rate == 0 and "Done." in lines[-1]
    per = out["periods"]
    assert len(per) == 2 and all(0.02 < p["recall20"] < 0.6 for p in per)


@pytest.mark.parametrize("H", [150, 64])
def test_batched_weight_gradient_products_match_float64(H):
    """ader_gemm_atb_x3_batch (the small-footprint form that runs inside the fused table update; H = 150 is the templated
    instantiation, 64 the generic one): dW = A^T . G and db = column sums of G for several products of ragged row counts,
    against float64 (bf16x3: <= 3e-5 of the tensor's max).
    Reference: the kernel gradients of modules.py:172-174 and :254-261."""
    import ctypes
    from ader_amd._lib import call, ptr
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    Ms = [1, 31, 32, 33, 1000, 2577]
    n = len(Ms)
    A = [torch.randn(m, H, generator=g).to(dev) for m in Ms]
    G = [torch.randn(m, H, generator=g).to(dev) for m in Ms]
    dW = [torch.full((H, H), float("nan"), device=dev) for _ in Ms]
    db = [torch.full((H,), float("nan"), device=dev) for _ in Ms]
    VP, IA = ctypes.c_void_p * n, ctypes.c_int * n
    Mi = IA(*Ms)
    slabs = call("ader_gemm_atb_batch_slabs", Mi, n)
    slab = torch.empty(slabs * 160 * 160, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call("ader_gemm_atb_x3_batch", VP(*[t.data_ptr() for t in A]), VP(*[t.data_ptr() for t in G]), VP(*[t.data_ptr() for t in dW]),
         VP(*[t.data_ptr() for t in db]), Mi, n, ptr(slab), H, st)
    torch.cuda.synchronize()
    for a, gg, w, b in zip(A, G, dW, db):
        want = a.double().t() @ gg.double()
        assert (w.double() - want).abs().max().item() <= 3e-5 * max(1.0, want.abs().max().item())
        wb = gg.double().sum(0)
        assert (b.double() - wb).abs().max().item() <= 3e-5 * max(1.0, wb.abs().max().item())
