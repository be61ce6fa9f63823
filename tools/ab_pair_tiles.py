"""A/B of the update kernel choice on small catalogs (dev tool): python tools/ab_pair_tiles.py cfgY|cfgD
k_tab32x3 (pairs of 64-row tiles) vs k_tab16x3 (single tiles) through ader_x3_update_pair_min_tiles; checks bit-identity first."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from ader_amd import _lib
from ader_amd.engine import Engine
from bench import synth_batch

w = sys.argv[1] if len(sys.argv) > 1 else "cfgY"
N, B, E = {"cfgY": (25750, 512, 102), "cfgD": (43105, 256, 143)}[w]
T = 50
batches = [synth_batch(B + E, T, N, 1000 * s, "cuda", "realistic") for s in range(4)]
teacher = torch.randn(E, int(0.9 * N), generator=torch.Generator().manual_seed(7)).cuda()
kw = dict(rate=0.3, teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device="cuda"), lambda_=0.8)
res = {}
for thr in (0, 768):
    _lib.load()
    _lib.call("ader_x3_update_pair_min_tiles", thr)
    eng = Engine(N, maxlen=T, seed=0)
    eng.pack_density, eng.pack_sessions = 0.1, True
    for i in range(30):
        sq, ps = batches[i % 4]
        eng.train_step(sq, ps[:B], N, 5e-4, **kw)
    torch.cuda.synchronize()
    res[thr] = eng.theta.clone()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(200):
        sq, ps = batches[i % 4]
        eng.train_step(sq, ps[:B], N, 5e-4, **kw)
    ev1.record()
    torch.cuda.synchronize()
    print(w, "pair_min_tiles", thr, "ms/step %.4f" % (ev0.elapsed_time(ev1) / 200))
print("bit-identical after 30 steps:", torch.equal(res[0], res[768]))
