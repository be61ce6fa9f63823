"""Is the fused update of a small catalog bound by its hottest tile (dev tool)?  Same step shape and session lengths, ids Zipf(1.05) vs uniform."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd.engine import Engine, SectionTimer
from bench import synth_batch
N, B, E, T = 25750, 512, 102, 50
for law in ("zipf", "uniform"):
    batches = [synth_batch(B + E, T, N, 1000 * s, "cpu", "realistic") for s in range(4)]
    if law == "uniform":
        rs = np.random.RandomState(5)
        nb = []
        for sq, ps in batches:
            sq = sq.numpy().copy(); ps = ps.numpy().copy()
            sq[sq != 0] = rs.randint(1, N + 1, size=int((sq != 0).sum()))
            ps[:] = rs.randint(1, N + 1, size=len(ps))
            nb.append((torch.from_numpy(sq), torch.from_numpy(ps)))
        batches = nb
    hot = max(int(np.bincount(((sq.numpy()[sq.numpy() != 0] - 1) // 64)).max()) for sq, _ in batches)
    batches = [(sq.cuda(), ps.cuda()) for sq, ps in batches]
    eng = Engine(N, maxlen=T)
    eng.pack_sessions, eng.pack_density = True, 0.1
    teacher = torch.randn(E, int(0.9 * N), generator=torch.Generator().manual_seed(7)).cuda()
    kw = dict(rate=0.3, teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device="cuda"), lambda_=0.8)
    for i in range(10):
        eng.train_step(batches[i % 4][0], batches[i % 4][1][:B], N, 5e-4, **kw)
    eng.timer = SectionTimer()
    for i in range(20):
        eng.train_step(batches[i % 4][0], batches[i % 4][1][:B], N, 5e-4, **kw)
    sec = eng.timer.collect()
    print(law, "entries in the hottest 64-item tile:", hot, {k: round(v, 4) for k, v in sec.items()})
