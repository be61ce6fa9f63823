"""Host enqueue time of one train step (no synchronisation inside the loop) vs its GPU time (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd.engine import Engine
from bench import synth_batch
# usage: cpu_overhead.py [cfgS|cfgY|cfgD] [pack: on|off] [driver: native|python]
w = sys.argv[1] if len(sys.argv) > 1 else "cfgS"
N, B, E, regime = {"cfgS": (1_000_000, 512, 0, "dense"), "cfgY": (25750, 512, 102, "realistic"), "cfgD": (43105, 256, 143, "realistic")}[w]
T = 50
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device=dev)
eng.pack_sessions = len(sys.argv) > 2 and sys.argv[2] == "on"
eng.pack_density = 0.1
eng.native_step = not (len(sys.argv) > 3 and sys.argv[3] == "python")
eng.warm_up()
batches = [synth_batch(B + E, T, N, 1000 * s, dev, regime) for s in range(4)]
kw = {}
if E:
    teacher = torch.randn(E, int(0.9 * N), generator=torch.Generator().manual_seed(7)).to(dev)
    kw = dict(teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device=dev), lambda_=0.8)
    batches = [(sq, ps[:B]) for sq, ps in batches]
_ts = eng.train_step
eng.train_step = lambda sq, ps, n, lr, rate: _ts(sq, ps, n, lr, rate=rate, **kw)
for i in range(8):
    eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
torch.cuda.synchronize()
for K in (4, 8):
    t0 = time.perf_counter()
    for i in range(K):
        eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("K=%d  host enqueue %.3f ms/step   total %.3f ms/step" % (K, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
print("plan hits %d, misses %d, errors %s" % (eng.plan_hits, eng.plan_misses, eng.plan_errors))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for i in range(200):
    eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(34); print(s.getvalue()[:7000])
