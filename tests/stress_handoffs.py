"""Stress of every in-launch hand-off between workgroups, on a COLD process (run by tests/test_gpu_handoffs.py in a fresh subprocess;
also `python tests/stress_handoffs.py [plans] [herdings]` by hand on the GPU box).

  1. ader_seq_pack_plan (csrc/seqp_plan.hip: per-workgroup session lengths handed to the last-arriving workgroup through a release
     fence + acq_rel ticket + acquire fence) -- `plans` launches over random (B, length law, window), the first launches of the
     process included, each compared with the numpy restatement of the packing rule below: every header word, every tile's row
     count, every session's first row and length, every packed row's records.
  2. ader_herding_select (csrc/herding.hip: a work ticket over a device-built list) -- `herdings` launches, all groups bit-equal to
     oracle/herding_ref (util.py:401-434 restated).
  3. one packed, distilled step of the YOOCHOOSE shape on four engines of this process (the first on cold memory): theta / Adam m /
     Adam v bit-identical between them.
Prints "handoffs ok ..." and exits 0, or raises."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from ader_amd import _lib  # noqa: E402


def numpy_plan(seq, T, window, row0=0, split=-1, row0_ex=0):
    """The packing rule of csrc/seqp_plan.hip restated: -> dict(hdr[4], tile_rows, srow0, slen, ids, lpos, gpos, info) (rows in use)."""
    B = seq.shape[0]
    nz = seq != 0
    ln = (T - np.where(nz.any(1), nz.argmax(1), T - 1)).astype(np.int64)
    w1_min, w1_max, target = window
    short = ln <= 16
    P1 = int(ln[short].sum())
    w1 = w1_max
    if target > 0:
        w1 = max(w1_min, min(w1_max, (P1 + target - 1) // target))
    w1 = max(16, min(49, w1))
    n1 = (P1 - 1) // w1 + 1 if P1 > 0 else 0
    st = np.zeros(B, dtype=np.int64)
    st[short] = np.cumsum(ln[short]) - ln[short]
    raw = np.where(short, st // w1, n1 + np.cumsum(~short) - 1)
    nt_raw = n1 + int((~short).sum())
    first = np.full(nt_raw, np.iinfo(np.int64).max)
    end = np.zeros(nt_raw, dtype=np.int64)
    np.minimum.at(first, raw, st)
    np.maximum.at(end, raw, st + ln)
    used = end > 0
    compact = np.cumsum(used) - 1
    tile_rows = (end - first)[used]
    srow0 = compact[raw] * 64 + st - first[raw]
    nt = int(used.sum())
    n_rows = nt * 64
    ids = np.full(n_rows, -1, dtype=np.int64)
    lpos, gpos, info = ids.copy(), ids.copy(), ids.copy()
    b_idx = np.repeat(np.arange(B), ln)
    k = np.arange(int(ln.sum())) - np.repeat(np.cumsum(ln) - ln, ln)
    t = T - ln[b_idx] + k
    pr = srow0[b_idx] + k
    gb = np.where((split >= 0) & (b_idx >= split), b_idx - split + row0_ex, b_idx + row0)
    ids[pr] = seq[b_idx, t]
    lpos[pr] = b_idx * T + t
    gpos[pr] = gb * T + t
    info[pr] = (srow0[b_idx] & 63) | np.where(k == ln[b_idx] - 1, 64, 0) | (t << 8) | (b_idx << 16)
    return dict(hdr=np.array([nt, nt * 64, int(ln.sum()), w1]), tile_rows=tile_rows, srow0=srow0, slen=ln, ids=ids, lpos=lpos, gpos=gpos,
                info=info)


def random_batch(rs, T):
    B = int(rs.choice([1, 2, 63, 64, 65, 130, 256, 512, 614, 1000, 1024, int(rs.randint(1, 1500))]))
    law = rs.randint(4)
    if law == 0:
        ln = np.clip(rs.geometric(0.2, size=B), 1, T)
    elif law == 1:
        ln = rs.randint(0, T + 1, size=B)                      # (0: an all-padding session)
    elif law == 2:
        ln = np.clip(rs.geometric(0.35, size=B), 1, T)
        ln[rs.randint(0, B, size=max(1, B // 40))] = rs.choice([T, 33, 32, 17, 16])
    else:
        ln = np.full(B, int(rs.choice([1, 16, 17, T])))
    seq = np.zeros((B, T), dtype=np.int32)
    for b in np.flatnonzero(ln):
        seq[b, T - ln[b]:] = rs.randint(1, 5000, size=ln[b])
    window = [(17, 49, 224), (1, 49, 224), (49, 49, 0), (5, 5, 0), (17, 49, 1000)][rs.randint(5)]
    return seq, window


def stress_plans(n_launch, T=50, seed=0):
    dev = torch.device("cuda:0")
    i32 = dict(dtype=torch.int32, device=dev)
    cap = 1536
    bufs = {k: torch.zeros(n, **i32) for k, n in (("hdr", 8), ("tile_rows", cap), ("ids", cap * 64), ("lpos", cap * 64), ("gpos", cap * 64),
                                                  ("info", cap * 64), ("srow0", cap), ("slen", cap))}
    c = _lib.AderSeqPack()
    for k, t in bufs.items():
        setattr(c, k, t.data_ptr())
    import ctypes
    rs = np.random.RandomState(seed)
    st = torch.cuda.current_stream().cuda_stream
    pend = []
    for it in range(n_launch):
        seq, window = random_batch(rs, T)
        B = seq.shape[0]
        split, row0, row0_ex = (-1, 0, 0) if it % 3 else (B // 2, 7, 900)
        seq_d = torch.from_numpy(seq).to(dev)
        _lib.call("ader_seq_pack_plan", seq_d.data_ptr(), B, T, row0, split, row0_ex, *window, ctypes.byref(c), st)
        got = {k: bufs[k].cpu().numpy().astype(np.int64) for k in bufs}           # (stream-ordered copies: also the synchronisation)
        want = numpy_plan(seq, T, window, row0, split, row0_ex)
        nt = int(want["hdr"][0])
        assert got["hdr"][:4].tolist() == want["hdr"].tolist() and got["hdr"][7] == 0, (it, B, window, got["hdr"], want["hdr"])
        assert np.array_equal(got["tile_rows"][:nt], want["tile_rows"]), (it, B, window)
        assert np.array_equal(got["srow0"][:B], want["srow0"]) and np.array_equal(got["slen"][:B], want["slen"]), (it, B, window)
        live = want["ids"] >= 0
        for k in ("ids", "lpos", "gpos", "info"):
            assert np.array_equal(got[k][:nt * 64][live], want[k][live]), (it, B, window, k)
        pend.append(seq_d)
        if len(pend) > 4:
            pend.pop(0)
    return n_launch


def stress_herding(n_launch, seed=1):
    from make_golden import herding_inputs
    from oracle import herding_ref
    from ader_amd.exemplar import herding_max_steps
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(seed)
    shapes = [(64, 64, False), (65, 30, False), (129, 129, True), (511, 40, False), (513, 77, True), (3, 2, False), (20, 5, True), (1, 1, False),
              (752, 90, False), (40, 12, False), (2, 0, False), (300, 25, True)]
    reps = [herding_inputs(700 + i, nn, 150, dup) for i, (nn, mm, dup) in enumerate(shapes)]
    oracle = [herding_ref.herding_select(r, m)[0] for r, (_, m, _) in zip(reps, shapes)]
    H = 150
    for it in range(n_launch):
        order = rs.permutation(len(shapes))[:int(rs.randint(1, len(shapes) + 1))]
        rep = torch.from_numpy(np.concatenate([reps[g] for g in order])).to(dev)
        offs = np.concatenate([[0], np.cumsum([shapes[g][0] for g in order])])
        quota = [min(shapes[g][1], shapes[g][0]) for g in order]
        n, G = rep.shape[0], len(order)
        seg = torch.tensor(offs, dtype=torch.int64, device=dev)
        q = torch.tensor(quota, dtype=torch.int32, device=dev)
        ms = torch.tensor([herding_max_steps(m) for m in quota], dtype=torch.int32, device=dev)
        D = torch.full((n * H + G + 64,), float("nan"), device=dev)
        chosen = torch.empty(n, dtype=torch.uint8, device=dev)
        sel = torch.zeros(n, dtype=torch.int32, device=dev)
        cnt = torch.full((G,), -1, dtype=torch.int32, device=dev)
        _lib.call("ader_herding_select", rep.data_ptr(), seg.data_ptr(), q.data_ptr(), ms.data_ptr(), G, n, H, D.data_ptr(), chosen.data_ptr(),
                  sel.data_ptr(), cnt.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        sel_h, cnt_h = sel.cpu().numpy(), cnt.cpu().numpy()
        for j, g in enumerate(order):
            assert sel_h[offs[j]:offs[j] + cnt_h[j]].tolist() == oracle[g], (it, j, shapes[g])
    return n_launch


def cold_warm_packed_step():
    """Four engines, the first on the cold process: two packed distilled steps of the YOOCHOOSE shape, bitwise the same state."""
    import bench
    from ader_amd.engine import Engine
    _, N, B, E = bench.REAL_SHAPES["cfgY"]
    T = 50
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    teacher = torch.randn(E, int(0.9 * N), generator=g).to(dev)
    batches = [bench.synth_batch(B + E, T, N, 100 + s, dev, "realistic") for s in range(2)]
    out = []
    for _ in range(4):
        eng = Engine(N, maxlen=T, seed=0, device=dev)
        eng.pack_sessions = True
        for seq, pos in batches:
            eng.train_step(seq, pos[:B], N, 5e-4, rate=0.3, teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device=dev), lambda_=0.8)
        torch.cuda.synchronize()
        eng.check_status()
        out.append((eng.theta.clone(), eng.adam_m.clone(), eng.adam_v.clone(), float(eng.loss)))
        del eng
        torch.cuda.empty_cache()
    for k in range(1, 4):
        assert out[k][3] == out[0][3], (k, out[k][3], out[0][3])
        for a, b, name in zip(out[0][:3], out[k][:3], ("theta", "m", "v")):
            assert torch.equal(a, b), "engine %d vs 0: %s differs in %d elements" % (k, name, int((a != b).sum()))
    return 4


if __name__ == "__main__":
    n_plans = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    n_herd = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    a = stress_plans(n_plans)                  # FIRST: the plan launches are the first launches of this process
    b = stress_herding(n_herd)
    c = cold_warm_packed_step()
    print("handoffs ok: %d pack plans, %d herding launches, %d engines bit-identical" % (a, b, c))
