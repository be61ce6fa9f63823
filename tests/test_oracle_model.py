"""Self-checks of the model oracle (oracle/ader_ref_cpu.py).  The reference ships no tests or golden
vectors for the TF graph and TensorFlow cannot run here, so these are the hand-derived cases of
SURVEY section 4 plus fp64 finite differences -- they pin the restatement's internal consistency, not
TF output ("parity unpinned" for the model math)."""
import os

import pytest

import numpy as np
import torch

from oracle import ader_ref_cpu as R

T, H, L = 6, 8, 2


def small(dtype=torch.float64, seed=0, item_num=20):
    p = R.init_params(item_num, T, H, L, seed=seed, dtype=dtype)
    g = torch.Generator().manual_seed(seed + 1)
    for k in p:                                # non-trivial LN params / biases
        if k.endswith(("_b", "bq", "bk", "bv", "b1", "b2")):
            p[k] = (torch.randn(p[k].shape, generator=g, dtype=torch.float64) * 0.1).to(dtype)
        if k.endswith("_g"):
            p[k] = (1 + torch.randn(p[k].shape, generator=g, dtype=torch.float64) * 0.1).to(dtype)
    return p


def test_layernorm_of_zero_row_is_beta():
    g = torch.rand(H, dtype=torch.float64)
    b = torch.rand(H, dtype=torch.float64)
    out = R.layernorm(torch.zeros(3, H, dtype=torch.float64), g, b)
    assert torch.equal(out, b.expand(3, H))


def test_row0_embedding_is_zero_and_pads_do_not_leak():
    p = small()
    p["emb"][0] = 123.0                         # must be ignored (modules.py:124-126)
    seq_a = torch.tensor([[0, 0, 0, 3, 4, 5]])
    rep_a = R.forward_rep(p, seq_a, L, 1)
    p["emb"][0] = -7.0
    rep_b = R.forward_rep(p, seq_a, L, 1)
    assert torch.equal(rep_a, rep_b)


def test_fully_masked_softmax_row_is_uniform():
    p = small()
    seq = torch.tensor([[0, 0, 0, 0, 2, 9]])
    _, inter = R.forward_rep(p, seq, 1, 1, return_intermediates=True)
    # pad query rows attend uniformly over all T keys (every score equals the padding constant);
    # their block output is re-zeroed by the mask (ADER.py:80)
    assert torch.all(inter["blk0"][0, :4] == 0)
    assert torch.all(inter["blk0"][0, 4:].abs().sum(-1) > 0)


def test_causality_last_position_only_sees_the_past():
    p = small()
    a = torch.tensor([[1, 2, 3, 4, 5, 6]])
    b = torch.tensor([[1, 2, 3, 4, 5, 7]])
    _, ia = R.forward_rep(p, a, L, 1, return_intermediates=True)
    _, ib = R.forward_rep(p, b, L, 1, return_intermediates=True)
    assert torch.equal(ia["final"][0, :5], ib["final"][0, :5])
    assert not torch.equal(ia["final"][0, 5], ib["final"][0, 5])


def test_loss_matches_manual_onehot_ce_and_kd():
    p = small()
    seq = torch.tensor([[0, 0, 1, 2, 3, 4], [0, 5, 6, 7, 8, 9], [0, 0, 0, 0, 10, 11]])
    pos = torch.tensor([5, 10])
    tl = torch.randn(1, 12, dtype=torch.float64)
    lam = 0.37
    loss = R.loss_fn(p, seq, pos, 15, L, 1, ex_logits=tl, lambda_=lam, training=False)
    rep = R.forward_rep(p, seq, L, 1)
    lg = R.logits_from_rep(p, rep, 15)
    ce = torch.stack([-torch.log_softmax(lg[i], -1)[pos[i] - 1] for i in range(2)]).mean()
    kd = -(torch.softmax(tl[0], -1) * torch.log_softmax(lg[2, :12], -1)).sum()
    assert torch.allclose(loss, ce + lam * kd, rtol=1e-12, atol=1e-12)


def test_heads_split_on_channels():
    p = small()
    seq = torch.tensor([[0, 1, 2, 3, 4, 5]])
    r1 = R.forward_rep(p, seq, L, 1)
    r2 = R.forward_rep(p, seq, L, 2)
    assert not torch.allclose(r1, r2)


def test_gradients_by_finite_differences_fp64():
    p = small()
    seq = torch.tensor([[0, 0, 1, 2, 3, 4], [6, 5, 6, 7, 8, 9], [0, 0, 0, 0, 10, 11]])
    pos = torch.tensor([5, 10])
    tl = torch.randn(1, 12, dtype=torch.float64)
    kw = dict(ex_logits=tl, lambda_=0.5, training=True, rate=0.25, seed=3, step=7)
    loss, grads = R.loss_and_grads(p, seq, pos, 15, L, 1, **kw)
    rs = np.random.RandomState(0)
    for name in ("emb", "pos", "b0.wq", "b0.wk", "b1.wv", "b1.w1", "b0.w2", "b0.ln1_g", "b1.ln2_b", "lnf_g", "b0.bq", "b1.b2"):
        flat = p[name].view(-1)
        for _ in range(3):
            i = int(rs.randint(0, flat.numel()))
            if name == "emb" and i < H:          # row 0 never receives gradient
                continue
            old = flat[i].item()
            eps = 1e-6
            flat[i] = old + eps
            lp = R.loss_fn(p, seq, pos, 15, L, 1, **kw).item()
            flat[i] = old - eps
            lm = R.loss_fn(p, seq, pos, 15, L, 1, **kw).item()
            flat[i] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - grads[name].view(-1)[i].item()) < 1e-6 * max(1.0, abs(fd)), (name, i)
    assert torch.all(grads["emb"][0] == 0)


def test_dropout_mask_spec_statistics_and_determinism():
    k1 = R.dropout_keep(200000, 0, 0, 5, 1, 0.3)
    k2 = R.dropout_keep(200000, 0, 0, 5, 1, 0.3)
    k3 = R.dropout_keep(200000, 0, 0, 6, 1, 0.3)
    assert np.array_equal(k1, k2) and not np.array_equal(k1, k3)
    assert abs(k1.mean() - 0.7) < 0.005
    # a shard starting at element 1000 sees the same mask as the slice of the full tensor
    assert np.array_equal(R.dropout_keep(500, 1000, 0, 5, 1, 0.3), k1[1000:1500])


def test_tf_adam_first_step_moves_by_lr():
    p = {"w": torch.tensor([1.0, -2.0], dtype=torch.float32)}
    opt = R.TFAdam(p)
    opt.step(p, {"w": torch.tensor([0.5, -3.0])}, 1e-3)
    # t=1: m = .1 g, v = .001 g^2, lr_t = lr*sqrt(.001)/.1  ->  step = lr * g/|g| (up to eps)
    assert torch.allclose(p["w"], torch.tensor([1.0 - 1e-3, -2.0 + 1e-3]), atol=1e-7)


def test_rank_matches_double_argsort():
    p = small(dtype=torch.float32)
    seq = torch.tensor([[0, 0, 1, 2, 3, 4], [6, 5, 6, 7, 8, 9]])
    ranks = R.rank_all(p, seq, 20, L, 1).numpy()
    rep = R.forward_rep(p, seq, L, 1)
    lg = R.logits_from_rep(p, rep, 20).detach().numpy()
    for b in range(2):
        for t in (1, 7, 20):
            assert R.rank_of_target(lg[b], t) == ranks[b, t - 1]
    assert R.metrics([0, 9, 10, 19, 20]) == (sum(1 / (r + 1) for r in (0, 9, 10, 19)) / 5, 4 / 5, (1 + 0.1) / 5, 2 / 5)


def test_oracle_trained_on_period_1_lands_on_the_published_curve(golden_dir):
    """The only reference-held values the ORACLE ITSELF can be checked against: the figure's period-1 points (results.svg ->
    results_svg_curves.json).  tests/golden/make_oracle_period1.py trained oracle/ader_ref_cpu.py (torch-CPU float32, TF-Adam, the
    reference's flags and early stopping, the golden-pinned host feeders) on DIGINETICA period 1 in the build container (13 CPU
    minutes) and recorded its test metrics; period 1 of ADER, Dropout and Joint is one and the same configuration (vanilla loss,
    dropout 0.3), i.e. three independent runs of the reference: 49.45 / 49.17 / 49.21 Recall@20, 17.71 / 17.63 / 17.66 MRR@20.
    The oracle must land inside that cloud (+- 0.35 point around its mean: the reference's own spread is 0.28)."""
    import json
    rec = json.load(open(os.path.join(golden_dir, "oracle_period1.json")))["default"]
    curves = json.load(open(os.path.join(golden_dir, "results_svg_curves.json")))["curves"]["DIGINETICA"]
    r_ref = [curves[m]["recall20"][0] for m in ("ADER", "Dropout", "Joint")]
    m_ref = [curves[m]["mrr20"][0] for m in ("ADER", "Dropout", "Joint")]
    r20, m20 = 100.0 * rec["test"]["recall20"], 100.0 * rec["test"]["mrr20"]
    assert rec["max_item"] == 18569 and rec["batch_num"] == 196 and rec["steps"] == rec["epochs_run"] * 196
    assert abs(r20 - sum(r_ref) / 3) <= 0.35, (r20, r_ref)
    assert abs(m20 - sum(m_ref) / 3) <= 0.35, (m20, m_ref)
    # the validation curve rises monotonically to its best epoch and early stopping ran its 5 epochs of patience
    v = [e["valid_recall20"] for e in rec["valid_log"]]
    assert v.index(max(v)) + 1 == rec["best_epoch"] and rec["epochs_run"] == rec["best_epoch"] + 5
    # second configuration: dropout 0 (period 1 of the figure's Finetune and EWC curves: 46.31 / 46.19 Recall@20, 16.50 / 16.39 MRR)
    ft = json.load(open(os.path.join(golden_dir, "oracle_period1.json")))["finetune"]
    r_ft = [curves[m]["recall20"][0] for m in ("Finetune", "EWC")]
    m_ft = [curves[m]["mrr20"][0] for m in ("Finetune", "EWC")]
    assert ft["steps"] == ft["epochs_run"] * 196 and ft["epochs_run"] == ft["best_epoch"] + 5
    assert abs(100.0 * ft["test"]["recall20"] - sum(r_ft) / 2) <= 0.35, (ft["test"], r_ft)
    assert abs(100.0 * ft["test"]["mrr20"] - sum(m_ft) / 2) <= 0.35, (ft["test"], m_ft)
    # and the oracle reproduces the reference's dropout effect at period 1 (about +3 points of Recall@20)
    assert 2.0 <= r20 - 100.0 * ft["test"]["recall20"] <= 4.0


def test_oracle_finetune_16_periods_follow_the_published_curve(golden_dir):
    """The oracle through the reference's WHOLE continual loop: tests/golden/make_oracle_finetune16.py ran oracle/ader_ref_cpu.py
    (torch-CPU float32, TF-Adam) as the Finetune baseline (main.py:141-146: dropout 0, no exemplars; previous best state restored per
    period, early stopping, test on the next period) over the 16 DIGINETICA periods in the build container (66 CPU-minutes) and recorded
    the per-period test metrics.  Against the Finetune curve of the reference's published figure (results.svg): 16-period averages
    47.53 / 16.18 against 47.28 / 16.01, per-period mean |delta| 0.43 / 0.23 (largest 1.22 / 0.54) -- the spread between two runs of the
    reference itself (its period-1 points of one configuration differ by 0.28).  The HIP engine's run of the same configuration is
    compared with these 16 values in tests/test_gpu_e2e_parity.py."""
    import json
    rec = json.load(open(os.path.join(golden_dir, "oracle_finetune16.json")))
    ref = json.load(open(os.path.join(golden_dir, "results_svg_curves.json")))["curves"]["DIGINETICA"]["Finetune"]
    per = rec["periods"]
    assert [p["period"] for p in per] == list(range(1, 17)) and per[-1]["max_item"] == 43105
    assert all(p["epochs_run"] == p["best_epoch"] + 5 for p in per)          # early stopping ran its patience in every period
    for key, tol_avg, tol_mad in (("recall20", 0.5, 0.6), ("mrr20", 0.35, 0.35)):
        mine = [100.0 * p[key] for p in per]
        avg, avg_ref = sum(mine) / 16, sum(ref[key]) / 16
        mad = sum(abs(a - b) for a, b in zip(mine, ref[key])) / 16
        assert abs(avg - avg_ref) <= tol_avg, (key, avg, avg_ref)
        assert mad <= tol_mad, (key, mad)


def test_oracle_ader_16_periods_follow_the_published_curve(golden_dir):
    """The METHOD through the oracle: tests/golden/make_oracle_ader16.py ran ADER with its default flags (herding exemplars, adaptive
    distillation, dropout 0.3) over the 16 DIGINETICA periods on oracle/ader_ref_cpu.py + oracle/herding_ref.py, driven by the product's
    host loop with the HIP model swapped for an oracle-backed stand-in (no HIP kernel runs; 239 CPU-minutes).  Against the ADER curve of the
    reference's published figure: 16-period averages 50.28 / 17.42 against 50.21 / 17.32, per-period mean |delta| 0.20 / 0.12 (largest
    0.69 / 0.26) -- the oracle, its distillation loss and the herding restatement included, reproduces the published run period by period."""
    import json
    path = os.path.join(golden_dir, "oracle_ader16.json")
    rec = json.load(open(path))
    ref = json.load(open(os.path.join(golden_dir, "results_svg_curves.json")))["curves"]["DIGINETICA"]["ADER"]
    per = rec["periods"]
    assert [p["period"] for p in per] == list(range(1, 17)) and per[-1]["max_item"] == 43105
    for key, tol_avg, tol_mad in (("recall20", 0.3, 0.4), ("mrr20", 0.25, 0.25)):
        mine = [100.0 * p[key] for p in per]
        avg, avg_ref = sum(mine) / 16, sum(ref[key]) / 16
        mad = sum(abs(a - b) for a, b in zip(mine, ref[key])) / 16
        assert abs(avg - avg_ref) <= tol_avg, (key, avg, avg_ref)
        assert mad <= tol_mad, (key, mad)


def test_oracle_er_herding_record_is_the_reference_flag_set():
    """tests/golden/oracle_er4.json: the oracle through four DIGINETICA periods of the poster's ER-herding column at the reference's
    documented command line (one-hot replay, base weight 0.8).  The `-m gpu` suite holds the HIP engine to these values
    (tests/test_gpu_e2e_parity.py::test_er_herding_follows_the_oracle); here: the record is what it says, and one-hot replay at this
    weight sits BELOW the distilled run of the same periods in the oracle too (the poster has ER 0.77 under ADER over 16 periods)."""
    import json
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    er = json.load(open(os.path.join(g, "oracle_er4.json")))
    ad = json.load(open(os.path.join(g, "oracle_ader16.json")))
    assert "--disable_distillation True" in er["config"] and len(er["periods"]) == 4 and "not bitwise reproducible" in er["reproducibility"]
    assert [p["max_item"] for p in er["periods"]] == [p["max_item"] for p in ad["periods"][:4]]
    r_er = [100 * p["recall20"] for p in er["periods"]]
    r_ad = [100 * p["recall20"] for p in ad["periods"][:4]]
    assert abs(r_er[0] - r_ad[0]) < 0.5                                  # period 1 has no exemplars: the same run up to thread noise
    assert all(a - e > 0.5 for e, a in zip(r_er[1:], r_ad[1:]))           # replay < distillation from period 2 on
