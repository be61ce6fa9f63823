"""Packed session tiles (csrc/seqp_*.hip) against the unpacked session kernels and the CPU oracle.

The reference left-pads every session to maxlen (util.py:161-169) and computes the padding (ADER.py:41-91); the packed kernels
drop it -- exact, because a padded position influences no real one (modules.py:188-193, ADER.py:80).  These tests hold the packed
path to the unpacked kernels on every saved activation and every gradient (same arithmetic; only summation orders differ: a
session's keys sit at other tile rows, LayerNorm partials are per tile) and to the fp64 oracle at the bounds of test_gpu_parity.py.
The plan itself (index work) is checked exactly."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))
from test_gpu_parity import R, _engine, _params, nerr  # noqa: E402


def _law(rs, B, T, n_items, law):
    """Session-length laws: 'geom' = 1 + Geometric(0.2) clipped (the shipped splits' shape, bench.py --regime realistic),
    'mixed' = short sessions with a maxlen session, a 33-, a 32-, a 17- and a 16-item one beside them, 'uniform' = U[1,T]."""
    seq = np.zeros((B, T), dtype=np.int32)
    if law == "geom":
        ln = np.clip(rs.geometric(0.2, size=B), 1, T)
    elif law == "uniform":
        ln = rs.randint(1, T + 1, size=B)
    else:
        ln = np.clip(rs.geometric(0.35, size=B), 1, T)
        special = [T, min(T, 33), min(T, 32), min(T, 17), min(T, 16), 1, 1, T]
        for i, v in enumerate(special):
            ln[(3 + 7 * i) % B] = v
    for b in range(B):
        seq[b, T - ln[b]:] = rs.randint(1, n_items + 1, size=ln[b])
    return seq


def _plan(eng, seq_d, window=None):
    if window is not None:
        eng.pack_window = window
    return eng._pack_plan(seq_d, "tst")


@pytest.mark.parametrize("law", ["geom", "mixed", "uniform"])
@pytest.mark.parametrize("B", [1, 7, 130, 614, 1024])
@pytest.mark.parametrize("window", [(17, 49, 224), (1, 49, 224), (49, 49, 0), (5, 5, 0)])
def test_pack_plan_is_a_tiling_of_the_real_positions(law, B, window):
    T = 50
    eng = _engine(300, T, 150, 2, 1)
    rs = np.random.RandomState(B)
    seq = _law(rs, B, T, 300, law)
    if B > 3:
        seq[2] = 0                      # an all-padding session keeps position T-1
    pk = _plan(eng, eng._dev_i32(seq), window)
    torch.cuda.synchronize()
    hdr = pk["hdr"].cpu().numpy()
    nt = int(hdr[0])
    tr = pk["trows"].cpu().numpy()[:nt]
    lpos = pk["lpos"].cpu().numpy()
    ids = pk["ids"].cpu().numpy()
    info = pk["info"].cpu().numpy()
    gpos = pk["gpos"].cpu().numpy()
    srow0, slen = pk["srow0"].cpu().numpy(), pk["slen"].cpu().numpy()
    ln = T - np.where((seq != 0).any(1), (seq != 0).argmax(1), T - 1)        # first item .. T-1; an all-padding session: position T-1
    assert hdr[1] == 64 * nt and (tr >= 1).all() and (tr <= 64).all() and nt <= B
    assert np.array_equal(slen, ln) and hdr[2] == ln.sum() == tr.sum()
    seen = np.zeros(B * T, dtype=np.int64)
    used = np.zeros(nt * 64, dtype=np.int64)
    prev_tile_b = {}
    for b in range(B):
        p0, n = int(srow0[b]), int(ln[b])
        tile, r0 = divmod(p0, 64)
        assert tile < nt and r0 + n <= tr[tile]                       # a session never leaves its tile
        assert prev_tile_b.get(tile, -1) < b                          # batch order inside a tile
        prev_tile_b[tile] = b
        rows = np.arange(p0, p0 + n)
        used[rows] += 1
        t = T - n + np.arange(n)
        assert np.array_equal(lpos[rows], b * T + t) and np.array_equal(ids[rows], seq[b, t])
        assert np.array_equal(gpos[rows], b * T + t)
        assert np.array_equal(info[rows] & 63, np.full(n, r0)) and np.array_equal((info[rows] >> 8) & 63, t)
        assert np.array_equal(info[rows] >> 16, np.full(n, b))
        assert np.array_equal((info[rows] >> 6) & 1, (np.arange(n) == n - 1).astype(np.int64))
        seen[b * T + t] += 1
    real = np.zeros((B, T), dtype=bool)
    for b in range(B):
        real[b, T - ln[b]:] = True
    assert np.array_equal(seen.reshape(B, T) == 1, real) and seen.max() == 1
    for u in range(nt):
        assert (used[u * 64:u * 64 + tr[u]] == 1).all() and (used[u * 64 + tr[u]:(u + 1) * 64] == 0).all()


def test_pack_plan_global_positions_of_a_data_parallel_shard():
    T = 50
    eng = _engine(300, T, 150, 2, 1)
    rs = np.random.RandomState(3)
    seq = _law(rs, 40, T, 300, "geom")
    eng.row0, eng.split_rows, eng.row0_ex = 120, 30, 1000
    pk = _plan(eng, eng._dev_i32(seq))
    torch.cuda.synchronize()
    srow0, slen, gpos = pk["srow0"].cpu().numpy(), pk["slen"].cpu().numpy(), pk["gpos"].cpu().numpy()
    for b in range(40):
        gb = 120 + b if b < 30 else 1000 + b - 30
        t = T - slen[b] + np.arange(slen[b])
        assert np.array_equal(gpos[srow0[b]:srow0[b] + slen[b]], gb * T + t)


def _pair(item_num, T, H, L, seed=0, **kw):
    a = _engine(item_num, T, H, L, 1, seed=seed, **kw)
    b = _engine(item_num, T, H, L, 1, seed=seed, **kw)
    a.pack_sessions, b.pack_sessions = False, True
    return a, b


ACT_KEYS = ["x", "q_in", "mean1", "std1", "kmask", "qmask", "Q", "K", "V", "x1", "y", "mean2", "std2", "h1d"]


@pytest.mark.parametrize("law", ["geom", "mixed"])
@pytest.mark.parametrize("rate", [0.0, 0.3])
@pytest.mark.parametrize("L", [1, 2])
@pytest.mark.parametrize("window", [(17, 49, 224), (49, 49, 0)])
def test_packed_forward_equals_unpacked_on_every_saved_activation(law, rate, L, window):
    """Tolerance 2e-5 (normalised max error) on every saved activation at the real positions and on rep: the two paths run the
    same arithmetic; the softmax denominator and P.V are summed in another order."""
    item_num, T, H, B = 700, 50, 150, 150
    eu, ep = _pair(item_num, T, H, L)
    ep.pack_window = window
    rs = np.random.RandomState(5)
    seq = _law(rs, B, T, 650, law)
    seq[4] = 0
    sd = eu._dev_i32(seq)
    ru = eu.forward(sd, training=rate > 0, rate=rate, step=7, save=True).clone()
    rp = ep.forward(ep._dev_i32(seq), training=rate > 0, rate=rate, step=7, save=True).clone()
    torch.cuda.synchronize()
    assert ep._act.get("pack") is not None and eu._act.get("pack") is None
    real = torch.from_numpy(seq != 0).reshape(-1)
    alive = torch.from_numpy((seq != 0).any(1))
    assert nerr(rp.cpu()[alive], ru.cpu()[alive]) < 2e-5
    assert nerr(rp.cpu(), ru.cpu()) < 2e-5             # the all-padding session too: rep = beta of the final LayerNorm
    for l in range(L):
        Su, Sp = eu._act[l], ep._act[l]
        for k in ACT_KEYS + ["x2"] * (l + 1 < L):
            u = Su[k] if k != "x2" else eu._act[l + 1]["x"]
            p = Sp[k] if k != "x2" else ep._act[l + 1]["x"]
            compact = Su["pruned"] and k not in ("x", "K", "V", "kmask")
            if compact:
                a, b = p.cpu()[alive], u.cpu()[alive]
            else:
                a, b = ep.unpack_rows(p).cpu()[real], u.cpu()[real]
            assert nerr(a, b, floor=1e-3) < 2e-5, (l, k)
    ep.check_status()


@pytest.mark.parametrize("mode", ["vanilla", "kd", "onehot_ex"])
@pytest.mark.parametrize("law", ["geom", "mixed"])
def test_packed_loss_and_gradients_equal_unpacked_and_match_the_oracle(mode, law):
    """Every gradient tensor packed vs unpacked <= 3e-5 (same arithmetic, other summation orders), and packed vs the fp64 oracle
    at the bounds of test_loss_and_gradients_match_oracle (6e-4 with the oracle following the device's ReLU branches)."""
    item_num, T, H, L, B, N = 700, 50, 150, 2, 140, 650
    eu, ep = _pair(item_num, T, H, L, seed=3, gemm="x3")
    rs = np.random.RandomState(2)
    seq = _law(rs, B, T, N, law)
    n_ex = 0 if mode == "vanilla" else B // 4
    n_tr = B - n_ex
    pos = rs.randint(1, N + 1, size=n_tr).astype(np.int32)
    kw = dict(rate=0.3)
    okw = {}
    if mode == "kd":
        Np = 600
        teacher = torch.randn(n_ex, Np, generator=torch.Generator().manual_seed(1)) * 2
        kw.update(teacher=teacher.cuda(), ex_trow=np.arange(n_ex, dtype=np.int32), lambda_=0.7)
        okw.update(ex_logits=teacher.double(), lambda_=0.7)
    elif mode == "onehot_ex":
        ex_pos = rs.randint(1, N + 1, size=n_ex).astype(np.int32)
        kw.update(ex_pos=ex_pos, lambda_=0.7)
        okw.update(ex_pos=ex_pos, lambda_=0.7)
    for e in (eu, ep):
        e.global_step = 4
        e.loss_and_grad(e._dev_i32(seq), pos, N, **kw)
    torch.cuda.synchronize()
    assert ep._act.get("pack") is not None
    assert abs(float(ep.loss) - float(eu.loss)) < 2e-6 * abs(float(eu.loss))
    for k in eu.layout:
        # (floor: the key bias has a true gradient of zero -- softmax is shift-invariant -- and holds float32 noise of ~1e-8)
        if k.endswith(".bk"):
            assert float((ep.gradient(k) - eu.gradient(k)).abs().max()) < 1e-6, k
            continue
        assert nerr(ep.gradient(k).cpu(), eu.gradient(k).cpu(), floor=1e-4) < 3e-5, k
    real = torch.from_numpy(seq != 0).reshape(-1)
    assert nerr(ep._last_g.cpu()[real], eu._last_g.cpu()[real]) < 3e-5
    masks = {}
    for l in range(L):          # the device's ReLU / dropout decisions, in the oracle's session-indexed layout
        S = ep._act[l]
        masks[l] = ("last", (S["h1d"] != 0).cpu()) if S["pruned"] else ("all", (ep.unpack_rows(S["h1d"]) != 0).cpu())
    oloss, og = R.loss_and_grads(_params(ep, torch.float64), seq, pos, N, L, 1, training=True, rate=0.3, seed=3, step=4,
                                 relu_masks=masks, **okw)
    assert abs(float(ep.loss) - float(oloss)) < 2e-5 * max(1.0, abs(float(oloss)))
    for k in ep.layout:
        assert nerr(ep.gradient(k).cpu().numpy(), og[k].numpy(), floor=1e-4) < 6e-4, k


def test_packed_train_steps_track_the_unpacked_path():
    """Three fused train steps (float32-grade logits, dropout on, exemplar rows distilled): parameters packed vs unpacked within
    1e-4 absolute -- a third of the bound test_x3_train_steps_track_the_oracle holds either path to against the oracle (Adam turns
    the float32 noise of a zero gradient, e.g. the key bias, into updates of the order of the learning rate times noise / eps)."""
    item_num, T, H, L, B, N = 900, 50, 150, 2, 200, 850
    eu, ep = _pair(item_num, T, H, L, seed=1, logits_dtype="x3")
    rs = np.random.RandomState(9)
    Np = 800
    teacher = (torch.randn(40, Np, generator=torch.Generator().manual_seed(2)) * 2).cuda()
    for step in range(3):
        seq = _law(rs, B, T, N, "geom" if step % 2 == 0 else "mixed")
        pos = rs.randint(1, N + 1, size=B - 40).astype(np.int32)
        for e in (eu, ep):
            e.train_step(e._dev_i32(seq), pos, N, 5e-4, rate=0.3, teacher=teacher, ex_trow=np.arange(40, dtype=np.int32), lambda_=0.8)
    torch.cuda.synchronize()
    assert ep._act.get("pack") is not None
    for k in eu.layout:
        d = float((ep.param(k) - eu.param(k)).abs().max())
        assert d < 1e-4, (k, d)
    assert abs(float(ep.loss) - float(eu.loss)) < 1e-5 * abs(float(eu.loss))


def test_packed_step_is_bitwise_reproducible():
    item_num, T, H, L, B, N = 500, 50, 150, 2, 96, 450
    outs = []
    for _ in range(2):
        e = _engine(item_num, T, H, L, 1, seed=2, logits_dtype="x3")
        e.pack_sessions = True
        rs = np.random.RandomState(4)
        for step in range(3):
            seq = _law(rs, B, T, N, "geom")
            pos = rs.randint(1, N + 1, size=B).astype(np.int32)
            e.train_step(e._dev_i32(seq), pos, N, 1e-3, rate=0.3)
        torch.cuda.synchronize()
        outs.append(e.theta.clone())
    assert torch.equal(outs[0], outs[1])


def test_cached_descriptors_change_nothing():
    """Engine.cache_descriptors (the descriptors of a step shape built once, dropout keys and batch pointer refreshed per step)
    against descriptors rebuilt every step: bitwise the same parameters after ten steps that alternate between two batch shapes,
    distilled and plain steps, with an evaluation forward in between and dropout on; and the cache was actually used."""
    item_num, T, H, L, N, Np = 700, 50, 150, 2, 650, 600
    teacher = (torch.randn(30, Np, generator=torch.Generator().manual_seed(2)) * 2).cuda()
    outs, losses = [], []
    for cached in (True, False):
        e = _engine(item_num, T, H, L, 1, seed=3, logits_dtype="x3")
        e.pack_sessions, e.cache_descriptors = True, cached
        rs = np.random.RandomState(11)
        ls = []
        for step in range(10):
            B = (160, 96)[step % 2]
            seq = _law(rs, B, T, N, "geom" if step % 3 else "mixed")
            if step % 4 == 3:
                pos = rs.randint(1, N + 1, size=B).astype(np.int32)
                e.train_step(e._dev_i32(seq), pos, N, 1e-3, rate=0.3)
            else:
                pos = rs.randint(1, N + 1, size=B - 30).astype(np.int32)
                e.train_step(e._dev_i32(seq), pos, N, 1e-3, rate=0.3, teacher=teacher, ex_trow=np.arange(30, dtype=np.int32),
                             lambda_=0.7)
            ls.append(float(e.loss))
            if step == 4:
                e.encode(_law(rs, 64, T, N, "geom"))
        torch.cuda.synchronize()
        if cached:
            assert any(k[0] == "fwdp" for k in e._dcache) and any(k[0] == "bwdp" for k in e._dcache), list(e._dcache)
        else:
            assert not e._dcache
        outs.append(e.theta.clone())
        losses.append(ls)
        e.check_status()
    assert losses[0] == losses[1]
    assert torch.equal(outs[0], outs[1])


def test_auto_packing_follows_the_batch_density():
    e = _engine(300, 50, 150, 2, 1)
    assert e.pack_sessions == "auto"
    rs = np.random.RandomState(0)
    sparse, dense = _law(rs, 64, 50, 300, "geom"), rs.randint(1, 301, size=(64, 50)).astype(np.int32)
    e.encode(sparse)
    assert e._density_now < 0.2
    e.forward(e._seq_in(sparse), save=True)
    assert e._act.get("pack") is not None
    e.forward(e._seq_in(dense), save=True)
    assert e._act.get("pack") is None
    e.forward(e._seq_in(torch.from_numpy(sparse).cuda()), save=True)        # a device batch tells nothing: the feeder's announcement decides
    assert e._act.get("pack") is None
    e.pack_density = 0.1
    e.forward(e._seq_in(torch.from_numpy(sparse).cuda()), save=True)
    assert e._act.get("pack") is not None
