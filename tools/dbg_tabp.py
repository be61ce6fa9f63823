"""Compare k_tabp with k_tab32x3 after ONE step (dev tool): where do theta / m / v differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd import _lib
from ader_amd.engine import Engine
N, B, T, H = 300_037, 512, 50, 150
g = torch.Generator().manual_seed(5)
seq = torch.randint(1, N + 1, (B, T), generator=g, dtype=torch.int32).numpy()
pos = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32).numpy()
out = []
for mode in (1, 0):
    _lib.load(); _lib.call("ader_x3_update_pipelined", mode)
    eng = Engine(N, maxlen=T, hidden_units=H, num_blocks=2, num_heads=1, seed=0, logits_dtype="x3")
    eng.train_step(seq, pos, N, 5e-4, rate=0.3)
    torch.cuda.synchronize()
    out.append([eng.view(getattr(eng, b), "emb").clone() for b in ("theta", "adam_m", "adam_v")])
    del eng
for i, name in enumerate(("theta", "m", "v")):
    x, y = out[0][i], out[1][i]
    d = (x != y)
    print(name, "differing elements", int(d.sum()), "of", d.numel(), "rows", int(d.any(1).sum()), "max|d|", float((x - y).abs().max()),
          "max|y|", float(y.abs().max()))
    if d.any():
        rel = ((x - y).abs() / (y.abs() + 1e-30))
        print("   max rel", float(rel[d].max()), "median rel", float(rel[d].median()))
        cols = d.sum(0).cpu().numpy()
        print("   per-column diff counts (first 20):", cols[:20].tolist(), "... last 10:", cols[-10:].tolist())
        rows = d.any(1).nonzero().view(-1).cpu().numpy()
        print("   row % 128 histogram of differing rows (first 16 bins of 8):", np.bincount((rows - 1) % 128 // 8, minlength=16).tolist())
        print("   rows//128 (pairs) sample:", np.unique((rows - 1) // 128)[:20].tolist(), "n pairs", len(np.unique((rows - 1) // 128)))
        r0 = int(rows[0])
        print("   row", r0, "x", x[r0, :6].tolist(), "y", y[r0, :6].tolist())
