"""The reference's call surface over the HIP engine: `Ader`, `Session.run(fetches, feed_dict)`, `Saver`.
These read like the reference's own call sites: main.py:233-256 (train feed), util.py:452-455 (selection fetch),
ADER.py:140-150 / util.py:320-326 (predict), main.py:209-213,280-283 (saver)."""
import argparse
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ITEMS, T, H, N = 500, 20, 64, 430


def _args(**kw):
    d = dict(maxlen=T, hidden_units=H, l2_emb=0.0, random_seed=3, num_blocks=2, num_heads=1, dropout_rate=0.3,
             disable_distillation=False, logits_dtype="f32")
    d.update(kw)
    return argparse.Namespace(**d)


def _model(**kw):
    from ader_amd.model import Ader
    m = Ader(ITEMS, _args(**kw))
    g = torch.Generator().manual_seed(5)
    for k in m.engine.layout:          # away from the init symmetry (LayerNorm beta = 0 makes the query mask rounding noise)
        if k.endswith("_b"):
            m.engine.param(k).copy_(torch.randn(m.engine.layout[k][1], generator=g) * 0.1)
    m.engine.refresh_shadow()
    return m


def _same_update(a, b):
    """Two runs of one dense-path step agree up to the float-atomic order of the sparse table-gradient scatter (and Adam's
    eps-scale sensitivity for ~zero gradients, cf. test_gpu_parity.test_fused_table_adam_equals_unfused_step)."""
    d = (a.engine.theta - b.engine.theta).abs().cpu().numpy()
    return np.mean(d < 2e-6) > 0.999 and d.max() < 1.1e-3


def _batch(rs, B):
    seq = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        ln = rs.randint(1, T + 1)
        seq[b, T - ln:] = rs.randint(1, N + 1, size=ln)
    return seq, rs.randint(1, N + 1, size=B).astype(np.int32)


def test_train_op_feed_equals_fast_path():
    """sess.run(model.train_op, {...}) (main.py:233-256, vanilla loss) leaves the parameters of engine.train_step."""
    from ader_amd.model import Session
    rs = np.random.RandomState(0)
    seq, pos = _batch(rs, 33)
    a, b = _model(), _model()
    a.set_vanilla_loss()
    with Session(a) as sess:
        sess.run(a.train_op, feed_dict={a.input_seq: seq, a.pos: pos, a.is_training: True, a.max_item: N,
                                        a.dropout_rate: 0.3, a.lr: 5e-4})
    b.engine.train_step(seq, pos, N, 5e-4, rate=0.3)
    torch.cuda.synchronize()
    assert _same_update(a, b) and a.engine.global_step == b.engine.global_step == 1


def test_distilled_train_op_feed_equals_fast_path():
    """update_loss(lambda_) + exemplar_logits feed (main.py:196-201,241-248; rows appended after the train rows,
    main.py:229) == train_step(teacher=, ex_trow=)."""
    from ader_amd.model import Session
    rs = np.random.RandomState(1)
    seq, pos = _batch(rs, 21)
    ex_seq, _ = _batch(rs, 9)
    Np = 300
    teacher = rs.standard_normal((9, Np)).astype(np.float32)
    a, b = _model(), _model()
    a.update_loss(lambda_=0.6)
    b.update_loss(lambda_=0.6)
    with Session(a) as sess:
        sess.run(a.train_op, feed_dict={a.input_seq: np.concatenate([seq, ex_seq]), a.pos: pos, a.is_training: True,
                                        a.max_item: N, a.exemplar_logits: [row.tolist() for row in teacher],
                                        a.dropout_rate: 0.3, a.lr: 5e-4})
    b.train_step(np.concatenate([seq, ex_seq]), pos, N, 5e-4, 0.3, teacher=torch.from_numpy(teacher).cuda(),
                 ex_trow=np.arange(9, dtype=np.int32))
    torch.cuda.synchronize()
    assert _same_update(a, b)


def test_selection_fetch_and_predict_against_oracle():
    """sess.run([model.rep, model.logits], ...) (util.py:452-455) and model.predict (ADER.py:140-150): representation and
    logits against the CPU oracle; the rank matrix is argsort(argsort(-logits)) of the DEVICE logits (ties -> lower index)
    and pred[label-1] agrees with the count-greater kernel the Evaluator uses."""
    from oracle import ader_ref_cpu as R
    from ader_amd.model import Session
    rs = np.random.RandomState(2)
    seq, pos = _batch(rs, 17)
    m = _model()
    with Session(m) as sess:
        rep, logits = sess.run([m.rep, m.logits], feed_dict={m.input_seq: seq, m.dropout_rate: 0.3, m.max_item: N,
                                                             m.is_training: False})
        pred = m.predict(sess, seq, list(range(1, N + 1)))
    params = {k: v.double() for k, v in m.engine.export_params().items()}
    rep_o = R.forward_rep(params, seq.astype(np.int64), m.engine.L, m.engine.heads, training=False)
    lg_o = R.logits_from_rep(params, rep_o, N)
    assert rep.shape == (17, H) and logits.shape == (17, N) and pred.shape == (17, N) and pred.dtype == np.int32
    assert np.abs(rep - rep_o.numpy()).max() < 1e-4 * max(1.0, np.abs(rep_o.numpy()).max())
    assert np.abs(logits - lg_o.numpy()).max() < 1e-4 * max(1.0, np.abs(lg_o.numpy()).max())
    want = np.argsort(np.argsort(-logits, axis=1, kind="stable"), axis=1, kind="stable")
    assert np.array_equal(pred, want.astype(np.int32))
    fast = m.rank_targets(seq, pos, N)
    assert [int(x) for x in fast] == [int(pred[i, pos[i] - 1]) for i in range(17)]


def test_saver_round_trip_restores_parameters_and_optimizer_state():
    """saver.save / restore carry every global variable: parameters, Adam slots, beta powers, global_step (main.py:209-213)."""
    from ader_amd.model import Saver, Session
    rs = np.random.RandomState(3)
    seq, pos = _batch(rs, 12)
    m = _model()
    saver = Saver(m)
    with Session(m) as sess, tempfile.TemporaryDirectory() as d:
        m.engine.train_step(seq, pos, N, 5e-4, rate=0.3)
        path = saver.save(sess, os.path.join(d, "epoch=1.ckpt"))
        snap = (m.engine.theta.clone(), m.engine.adam_m.clone(), m.engine.adam_v.clone(), m.engine.global_step)
        m.engine.train_step(seq, pos, N, 5e-4, rate=0.3)
        assert not torch.equal(m.engine.theta, snap[0])
        saver.restore(sess, path)
        assert torch.equal(m.engine.theta, snap[0]) and torch.equal(m.engine.adam_m, snap[1])
        assert torch.equal(m.engine.adam_v, snap[2]) and m.engine.global_step == snap[3]
        m.engine.train_step(seq, pos, N, 5e-4, rate=0.3)      # the restored state steps like the original did
        first = m.engine.theta.clone()
        saver.restore(sess, path)
        m.engine.train_step(seq, pos, N, 5e-4, rate=0.3)
        d = (m.engine.theta - first).abs().cpu().numpy()
        assert np.mean(d < 2e-6) > 0.999 and d.max() < 1.1e-3


def test_device_feeder_yields_the_host_feeder_batches():
    """Sampler.to_device (GPU-resident feeder): same shuffled index stream, batches gathered on the device."""
    import random
    from ader_amd.data import Sampler
    rs = np.random.RandomState(4)
    sessions = [rs.randint(1, 90, size=rs.randint(2, 30)).tolist() for _ in range(57)]
    out = []
    for dev in (False, True):
        random.seed(11)
        smp = Sampler(sessions, T, 16)
        if dev:
            smp.to_device(torch.device("cuda"))
        got = []
        for _ in range(2 * smp.batch_num() + 1):             # across a reshuffle
            seq, pos = smp.next_batch()
            if dev:
                assert seq.is_cuda and seq.dtype == torch.int32 and seq.is_contiguous() and pos.is_contiguous()
                seq, pos = seq.cpu().numpy(), pos.cpu().numpy()
            got.append((seq.copy(), pos.copy()))
        out.append(got)
    for (sa, pa), (sb, pb) in zip(*out):
        assert np.array_equal(sa, sb) and np.array_equal(pa, pb)
