"""Full-size checks at BASELINE.json's synthetic workload (1M-item catalog, batch 512, seq 50): size-independent
properties (the oracle cannot run at this size in seconds) plus spot checks against dense float32 rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, B, T, H = 1_000_000, 512, 50, 150


@pytest.fixture(scope="module")
def batch():
    g = torch.Generator().manual_seed(0)
    seq = torch.randint(1, N + 1, (B, T), generator=g, dtype=torch.int32)
    seq[:64, :20] = 0                                   # some left padding
    seq[5, -4:] = seq[6, -1]                            # repeated ids
    pos = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32)
    return seq.numpy(), pos.numpy()


DTYPES = ["x3", "bf16"]        # "x3" = the float32-grade kernels bench.py's headline runs (k_lx3p, k_tab32x3); "bf16" = the companion


def _engine(dtype, **kw):
    from ader_amd.engine import Engine
    eng = Engine(N, maxlen=T, hidden_units=H, num_blocks=2, num_heads=1, seed=0, logits_dtype=dtype, **kw)
    g = torch.Generator().manual_seed(1)
    for k in eng.layout:                                 # LN beta away from 0 (see smoke())
        if k.endswith("_b"):
            eng.param(k).copy_(torch.randn(eng.layout[k][1], generator=g) * 0.1)
    eng.refresh_shadow()
    return eng


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_gradient_sums_to_zero_over_the_catalog(batch, dtype):
    """sum_n dlogit[b,n] = 0 for every row (softmax minus one-hot), so the column sums of the whole table gradient must
    equal the column sums of the sparse input-embedding rows alone: a checksum over all 10^6 x 150 entries."""
    seq, pos = batch
    eng = _engine(dtype)
    eng.loss_and_grad(seq, pos, N, rate=0.3)
    torch.cuda.synchronize()
    demb = eng.gradient("emb").double().sum(0).cpu().numpy()
    sparse = (eng._last_g.double().sum(0) * np.sqrt(np.float32(H)).item()).cpu().numpy()
    scale = np.abs(sparse).max()
    assert scale > 0
    # bf16 rounding of the 512 x 10^6 probabilities leaves |sum_n p - 1| ~ 1e-4 per row; float32 grade: an order less
    assert np.abs(demb - sparse).max() < (2e-3 if dtype == "bf16" else 2e-4) * scale
    assert float(eng.loss.item()) == pytest.approx(np.log(N), abs=0.5)      # near-uniform softmax at initialisation


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_and_unfused_table_updates_agree_at_full_size(batch, dtype):
    seq, pos = batch
    out = []
    for fuse in (True, False):
        eng = _engine(dtype)
        eng.fuse_adam = fuse
        eng.train_step(seq, pos, N, 5e-4, rate=0.3)
        torch.cuda.synchronize()
        out.append((eng.param("emb")[1:200001].cpu().numpy().copy(), eng.param("emb")[-100000:].cpu().numpy().copy(),
                    eng.view(eng.adam_v, "emb")[1:200001].cpu().numpy().copy()))
        del eng
        torch.cuda.empty_cache()
    for a, b in zip(out[0][:2], out[1][:2]):
        d = np.abs(a - b)
        assert np.mean(d < 2e-6) > 0.999 and d.max() < 1.1e-3           # see test_fused_table_adam_equals_unfused_step
    assert np.abs(out[0][2] - out[1][2]).max() <= 1e-4 * np.abs(out[1][2]).max()


@pytest.mark.parametrize("dtype", DTYPES)
def test_rank_and_lse_against_dense_float32_rows(batch, dtype):
    seq, pos = batch
    eng = _engine(dtype)
    rows = slice(0, 64)
    ranks = eng.rank_targets(seq[rows], pos[rows], N)
    lg = eng.logits(seq[rows], N)                                          # [64, 1M] float32, exact-f32 MFMA path
    tgt = torch.as_tensor(pos[rows].astype(np.int64), device=lg.device) - 1
    t = lg.gather(1, tgt[:, None])
    idx = torch.arange(N, device=lg.device)[None, :]
    exp = ((lg > t) | ((lg == t) & (idx < tgt[:, None]))).sum(1).cpu().numpy()
    assert np.array_equal(ranks, exp)                                      # bit-exact index work at full size
    eng.loss_and_grad(seq, pos, N, rate=0.0)
    lse_bf16 = eng._ws["lg_lse"][:64].cpu().numpy()
    lse_f32 = torch.logsumexp(lg.double(), 1).cpu().numpy()
    # bf16: operand rounding of the logits; x3: float32 grade, the loss bound of the exact-f32 kernels (2e-5 relative)
    assert np.abs(lse_bf16 - lse_f32).max() < (2e-2 if dtype == "bf16" else 2e-5 * np.abs(lse_f32).max())


def _fp64_step_reference(E, rep, lab, w, chunk=50_000):
    """float64 restatement of ADER.py:91-93 + its gradient at full size, from the parameters themselves: per-row log-sum-exp
    over the whole catalog, the loss, dRep = w (softmax . E - E[label]) -- chunked over the items (a [512, 50k] tile at a time)."""
    Bn = rep.shape[0]
    rep64 = rep.double()
    m = torch.full((Bn,), -1e300, dtype=torch.float64, device=rep.device)
    l = torch.zeros(Bn, dtype=torch.float64, device=rep.device)
    O = torch.zeros(Bn, H, dtype=torch.float64, device=rep.device)
    n_items = E.shape[0] - 1
    for s0 in range(1, n_items + 1, chunk):
        Ec = E[s0:min(n_items + 1, s0 + chunk)].double()
        S = rep64 @ Ec.t()
        mn = torch.maximum(m, S.max(1).values)
        sc = torch.exp(m - mn)
        P = torch.exp(S - mn[:, None])
        l = l * sc + P.sum(1)
        O = O * sc[:, None] + P @ Ec
        m = mn
    lse = m + torch.log(l)
    El = E[lab.long()].double()
    tl = (rep64 * El).sum(1)
    loss = (w.double() * (lse - tl)).sum()
    drep = w.double()[:, None] * (O / l[:, None] - El)
    return lse, loss, drep


# Bounds of the float32-grade (bf16x3) kernels at full size against fp64, normalised by the tensor's max.  Round 6: set from what the
# kernels MEASURE (printed by the test: lse 1.5e-7, loss 8e-8, dRep 6e-8, Adam m 3.6e-6, Adam v 2.1e-5, theta 1.6e-9 absolute) with
# a 5-15x margin -- rounds 2-5 held them to the exact-f32 kernels' written bounds (2e-5 / 2e-5 / 3e-4 / 3e-4 / 6e-4 / 2e-6), 20-300x
# looser than what the three-pass products deliver: at the level of a step's outputs "float32 grade" means float32 rounding
# (an fp32 log-sum-exp of ~13.8 has an ulp of 9.5e-7 = 7e-8 relative; profiles/ab_f16x3_forward_r6.txt: fp16 operand pieces do not
# move lse or loss at all and tighten dRep by 1.3-7x)
LSE_BOUND, LOSS_BOUND, DREP_BOUND, M_BOUND, V_BOUND, THETA_BOUND = 1e-6, 5e-7, 1e-6, 3e-5, 1e-4, 2e-8


@pytest.mark.parametrize("n_items", [N, N - 75])
def test_x3_headline_kernels_against_float64_at_full_size(batch, n_items):
    """The credited kernels AT the credited size (B = 512, N = 10^6: k_lx3p's 31,250 table blocks, k_tab32x3's 7,813 tile
    pairs, the last one half a pair; N - 75: a ragged tail tile): one fused train step of the float32-grade path against a
    float64 restatement computed from the same parameters and the device's own representation / input-gradient rows --
    per-row log-sum-exp, loss and dRep of ALL 512 rows at bounds set from the measured errors (lse 1e-6, loss 5e-7, dRep 1e-6
    normalised: float32 rounding), and theta / Adam m / Adam v of three 64-row table tiles (the first, a middle one, the tail tile) after the
    fused update against a dense float64 TF-Adam of those rows (ADER.py:91-96)."""
    seq, pos = batch
    pos = np.minimum(pos, n_items).astype(np.int32)
    seq = np.minimum(seq, n_items).astype(np.int32)
    pos[0] = n_items                                  # a label and an input position in the tail tile
    seq[7, -1] = n_items
    eng = _engine("x3")
    assert eng.lx3 and eng.x3_update == "tab16"
    E0 = eng.param("emb").detach().clone()            # [N+1, H] before the step
    lr = 5e-4
    loss = eng.train_step(seq, pos, n_items, lr, rate=0.3)
    torch.cuda.synchronize()
    eng.check_status()
    dev = E0.device
    rep = eng._act["rep"][:B].detach().clone()
    lab = torch.as_tensor(pos.astype(np.int64), device=dev)
    w = torch.full((B,), 1.0 / B, dtype=torch.float32, device=dev)
    lse64, loss64, drep64 = _fp64_step_reference(E0[:n_items + 1], rep, lab, w)
    lse_dev = eng._ws["lg_lse"][:B].double()
    drep_dev = eng._ws["drep"][:B].double()
    measured = {"lse": float((lse_dev - lse64).abs().max()) / float(lse64.abs().max()),
                "loss": abs(float(loss.item()) - float(loss64)) / abs(float(loss64)),
                "drep": float((drep_dev - drep64).abs().max()) / float(drep64.abs().max())}
    assert measured["lse"] < LSE_BOUND, measured
    assert measured["loss"] < LOSS_BOUND, measured
    assert measured["drep"] < DREP_BOUND, measured
    # table rows: dE[n] = sum_b w_b (p[b,n] - [label_b = n]) rep_b  +  sqrt(H) * sum_{(b,t): seq[b,t] = n} g[b,t]
    g_rows = eng._last_g.double()                                          # [B*T, H] masked / dropout-scaled input-gradient rows
    ids = torch.as_tensor(seq.reshape(-1).astype(np.int64), device=dev)
    last_tile0 = (n_items - 1) // 64 * 64 + 1
    tiles = [1, 64 * 7000 + 1, last_tile0]
    b1, b2, eps = 0.9, 0.999, 1e-8
    lr_t = lr * np.sqrt(1.0 - b2) / (1.0 - b1)
    for r0 in tiles:
        r1 = min(n_items + 1, r0 + 64)
        rows = torch.arange(r0, r1, device=dev)
        Er = E0[r0:r1].double()
        P = torch.exp(rep.double() @ Er.t() - lse64[:, None])              # [B, rows]
        g = (P * w.double()[:, None]).t() @ rep.double()
        hit = (lab[:, None] == rows[None, :]).double() * w.double()[:, None]
        g -= hit.t() @ rep.double()
        sel = (ids[:, None] == rows[None, :])
        g += np.sqrt(np.float32(H)).item() * (sel.double().t() @ g_rows)
        m64 = (1 - b1) * g
        v64 = (1 - b2) * g * g
        th64 = Er - lr_t * m64 / (v64.sqrt() + eps)
        m_dev = eng.view(eng.adam_m, "emb")[r0:r1].double()
        v_dev = eng.view(eng.adam_v, "emb")[r0:r1].double()
        th_dev = eng.param("emb")[r0:r1].double()
        measured["m@%d" % r0] = float((m_dev - m64).abs().max()) / float(m64.abs().max())
        measured["v@%d" % r0] = float((v_dev - v64).abs().max()) / float(v64.abs().max())
        measured["theta@%d" % r0] = float((th_dev - th64).abs().max())
        assert measured["m@%d" % r0] < M_BOUND and measured["v@%d" % r0] < V_BOUND, (r0, measured)
        assert measured["theta@%d" % r0] < THETA_BOUND, (r0, measured)
    print("full size, N = %d, measured against fp64: %s" % (n_items, {k: float("%.3g" % v) for k, v in measured.items()}))
    # rows past max_item are never touched
    if n_items < N:
        assert torch.equal(eng.param("emb")[n_items + 1:], E0[n_items + 1:])
        assert float(eng.view(eng.adam_v, "emb")[n_items + 1:].abs().max()) == 0.0


@pytest.mark.parametrize("n_items,rows", [(N, B), (300_037, 640)])
def test_full_size_update_is_bitwise_reproducible_between_engines(n_items, rows):
    """Four engines of one process, two fused steps each at the full catalog size: theta, Adam m and Adam v of the WHOLE table must be
    bit-identical between them -- the first engine runs on a cold process (first launches, fresh memory), the others on a warm one.
    (This comparison, then against the role-split kernel k_tabp, is what caught that kernel writing a few wrong vectors on a cold
    process in one run out of several; k_tabp was removed in round 5, the check stays for the kernel that remains.)  Batches with a
    hot item (a bucket of hundreds of sparse rows: the heavy path), repeated labels, left padding, a ragged tail tile."""
    from ader_amd.engine import Engine
    g = torch.Generator().manual_seed(5)
    batches = []
    for s in range(2):
        seq = torch.randint(1, n_items + 1, (rows, T), generator=g, dtype=torch.int32)
        seq[:50, :30] = 0
        seq[60:120, -3:] = 777                      # a hot item: ~180 sparse rows in one bucket
        seq[7, -1] = n_items
        pos = torch.randint(1, n_items + 1, (rows,), generator=g, dtype=torch.int32)
        pos[3] = pos[4] = pos[5]
        pos[0] = n_items
        batches.append((seq.numpy(), pos.numpy()))
    out = []
    for _ in range(4):
        eng = Engine(n_items + 50, maxlen=T, hidden_units=H, num_blocks=2, num_heads=1, seed=0, logits_dtype="x3")
        for seq, pos in batches:
            eng.train_step(seq, pos, n_items, 5e-4, rate=0.3)
        torch.cuda.synchronize()
        eng.check_status()
        out.append([eng.view(getattr(eng, b), "emb").clone() for b in ("theta", "adam_m", "adam_v")] + [float(eng.loss.item())])
        del eng
        torch.cuda.empty_cache()
    for k in range(1, 4):
        for x, y, name in zip(out[0][:3], out[k][:3], ("theta", "m", "v")):
            if not torch.equal(x, y):
                r = (x != y).any(1).nonzero().view(-1).tolist()
                raise AssertionError("engine %d vs 0, %s: %d rows differ; first %s; max |d| %.3e" % (k, name, len(r), r[:12], float((x - y).abs().max())))
        assert out[0][3] == out[k][3]
    assert float(out[0][2][1:n_items + 1].abs().max()) > 0 and float(out[0][2][n_items + 1:].abs().max()) == 0.0
