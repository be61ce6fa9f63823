"""The native step driver (one C call per train step, csrc/step_plan.hip + ader_amd/engine/plan.py) against the Python-driven step:
theta / Adam m / Adam v BITWISE equal after 50 steps in vanilla / distilled / one-hot-replay x packed / unpacked session kernels, with
batches whose real row counts wander (several plans), and every replayed step cross-checked slot by slot in verify mode.
Reference: main.py:220-256 (one sess.run(train_op) per step)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_packed import _law  # noqa: E402
from test_gpu_parity import _engine  # noqa: E402

ITEM, T, H, L, N, NP = 900, 50, 150, 2, 850, 800


def _batches(mode, rs, n_steps, B=96, E=24):
    """(seq, pos, kwargs) of n_steps steps: a few distinct (real train rows, real exemplar rows) pairs, padded to B / E rows."""
    out = []
    teacher = (torch.randn(40, NP, generator=torch.Generator().manual_seed(5)) * 2).cuda()
    for s in range(n_steps):
        n_t = (B, B - 3, B, B - 1)[s % 4]
        n_e = 0 if mode == "vanilla" else (E, E, E - 2)[s % 3]
        seq = _law(rs, B + (E if n_e else 0), T, N, "geom" if s % 3 else "mixed")
        seq[n_t:B] = 0
        pos = rs.randint(1, N + 1, size=B).astype(np.int32)
        pos[n_t:] = 0
        kw = dict(rate=0.3, n_train_global=n_t)
        if n_e:
            seq[B + n_e:] = 0
            kw.update(n_ex_global=n_e, lambda_=0.7)
            if mode == "kd":
                tr = rs.randint(0, 40, size=E).astype(np.int32)
                tr[n_e:] = -1
                kw.update(teacher=teacher, ex_trow=tr)
            else:
                ep = rs.randint(1, N + 1, size=E).astype(np.int32)
                ep[n_e:] = 0
                kw.update(ex_pos=ep)
        out.append((seq, pos, n_t, kw))
    return out


def _run(mode, packed, native, verify=False, n_steps=50):
    e = _engine(ITEM, T, H, L, 1, seed=3, logits_dtype="x3")
    e.pack_sessions, e.native_step, e.plan_verify = packed, native, verify
    rs = np.random.RandomState(17)
    losses = []
    for seq, pos, n_t, kw in _batches(mode, rs, n_steps):
        e.row0, e.row0_ex, e._ex_row0_set = 0, n_t, "n_ex_global" in kw       # (dist.DataParallel.set_rows: counters of the unpadded batch)
        seq_d, pos_d = torch.from_numpy(seq).cuda(), torch.from_numpy(pos).cuda()
        kw = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        e.train_step(seq_d, pos_d, N, 1e-3, **kw)
        losses.append(float(e.loss))
    torch.cuda.synchronize()
    e.check_status()
    return e, losses


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("mode", ["vanilla", "kd", "onehot"])
def test_native_steps_are_bitwise_the_python_driven_steps(mode, packed):
    ref, l_ref = _run(mode, packed, native=False)
    nat, l_nat = _run(mode, packed, native=True)
    assert ref.plan_hits == 0 and not nat.plan_errors, nat.plan_errors
    assert nat.plan_hits >= 30, (nat.plan_hits, nat.plan_misses)              # a handful of shapes, each recorded at most twice
    assert l_ref == l_nat
    for a, b, name in ((ref.theta, nat.theta, "theta"), (ref.adam_m, nat.adam_m, "m"), (ref.adam_v, nat.adam_v, "v")):
        assert torch.equal(a, b), name
    assert (ref.global_step, float(ref.b1p), float(ref.b2p)) == (nat.global_step, float(nat.b1p), float(nat.b2p))


@pytest.mark.parametrize("mode,packed", [("kd", True), ("vanilla", False), ("onehot", True)])
def test_every_replayable_step_matches_its_plan_slot_by_slot(mode, packed):
    """plan_verify: each step that WOULD be replayed runs through Python under the recorder and is compared with the plan's patched
    launch sequence (arguments and descriptor bytes): the plan key misses nothing that shapes a launch."""
    e, _ = _run(mode, packed, native=True, verify=True, n_steps=30)
    assert e.plan_verified >= 15 and not e.plan_errors, (e.plan_verified, e.plan_misses, e.plan_errors)


def test_device_fed_steps_equal_the_host_assembled_batches():
    """train_step_fed (the batch cut on the device by ader_feed_step from packed rows + epoch index slices) against train_step on the
    same batch assembled by torch: bitwise the same parameters, natively driven and Python-driven."""
    from ader_amd.data import pack_rows
    rs = np.random.RandomState(4)
    n_rows, n_ex_rows, Bt, Be = 700, 60, 64, 16
    sessions = [rs.randint(1, N + 1, size=int(k)).tolist() for k in np.clip(rs.geometric(0.25, size=n_rows) + 1, 2, 70)]
    ex_sessions = [rs.randint(1, NP + 1, size=int(k)).tolist() for k in np.clip(rs.geometric(0.25, size=n_ex_rows) + 1, 2, 70)]
    rows_t = torch.from_numpy(pack_rows(sessions, T)[0]).cuda()
    rows_e = torch.from_numpy(pack_rows(ex_sessions, T)[0]).cuda()
    teacher = (torch.randn(n_ex_rows, NP, generator=torch.Generator().manual_seed(9)) * 2).cuda()
    perm_t = torch.from_numpy(rs.permutation(n_rows)).cuda()
    perm_e = torch.from_numpy(rs.permutation(n_ex_rows)).cuda()
    thetas = []
    for how in ("fed-native", "fed-python", "host"):
        e = _engine(ITEM, T, H, L, 1, seed=3, logits_dtype="x3")
        e.pack_sessions, e.pack_density, e.native_step = "auto", 0.1, how != "fed-python"
        o_t = o_e = 0
        for s in range(24):
            n_t, n_e = (Bt, Bt - 2, Bt)[s % 3], (Be, Be - 1)[s % 2]
            if how.startswith("fed"):
                e.train_step_fed((rows_t, perm_t, o_t, n_t, Bt, rows_e, perm_e, o_e % 32, n_e, Be), N, 1e-3, 0.3, teacher=teacher, lambda_=0.6)
            else:
                it, ie = perm_t[o_t:o_t + n_t], perm_e[o_e % 32:o_e % 32 + n_e]
                seq = torch.zeros(Bt + Be, T, dtype=torch.int32, device="cuda")
                pos = torch.zeros(Bt, dtype=torch.int32, device="cuda")
                tr = torch.full((Be,), -1, dtype=torch.int32, device="cuda")
                seq[:n_t], pos[:n_t] = rows_t[it, :T], rows_t[it, T]
                seq[Bt:Bt + n_e], tr[:n_e] = rows_e[ie, :T], ie.to(torch.int32)
                e.row0, e.row0_ex, e._ex_row0_set = 0, n_t, True
                e.train_step(seq, pos, N, 1e-3, rate=0.3, teacher=teacher, ex_trow=tr, lambda_=0.6, n_train_global=n_t, n_ex_global=n_e)
            o_t = (o_t + n_t) % 600
            o_e += n_e
        torch.cuda.synchronize()
        e.check_status()
        if how == "fed-native":
            assert e.plan_hits >= 10 and not e.plan_errors, (e.plan_hits, e.plan_misses, e.plan_errors)
        thetas.append((e.theta.clone(), e.adam_m.clone(), e.adam_v.clone(), float(e.loss)))
    for other in thetas[1:]:
        assert other[3] == thetas[0][3]
        for a, b in zip(thetas[0][:3], other[:3]):
            assert torch.equal(a, b)
