"""Feasibility of hipGraph replay of the whole train step (TIMING ONLY: dropout keys and the Adam step size are frozen into the
captured launches here).  Eager vs replay, wall per step and host time per step, at the real-data step shapes and the headline.
usage: python tools/exp_graph_step.py [cfgY|cfgD|cfgS] [--serial]   (--serial: no side stream inside the captured step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from ader_amd.engine import Engine, side_stream

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfgY"
serial = "--serial" in sys.argv
dev = torch.device("cuda:0")
T = 50
if name == "cfgS":
    N, B, E = 1_000_000, 512, 0
    regime = "dense"
else:
    _, N, B, E = bench.REAL_SHAPES[name]
    regime = "realistic"
batches = [bench.synth_batch(B + E, T, N, 1000 * s + 77, dev, regime) for s in range(4)]
eng = Engine(N, maxlen=T, seed=0, device=dev)
eng.pack_density = 0.1 if regime == "realistic" else 1.0
kw = dict(rate=0.3)
if E:
    Np = int(0.9 * N)
    teacher = torch.empty(E, (Np + 3) // 4 * 4, device=dev)[:, :Np]
    teacher.copy_(torch.randn(E, Np, generator=torch.Generator().manual_seed(7)))
    kw.update(teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device=dev), lambda_=0.8)
if serial:
    eng.lists_side_stream = False
    eng.late_side_stream = False
s_seq = batches[0][0].clone()
s_pos = batches[0][1][:B].clone()


def eager(i):
    eng.train_step(batches[i % 4][0], batches[i % 4][1][:B], N, 5e-4, **kw)


def timed(fn, n):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3


n = 400 if name != "cfgS" else 100
print(name, "serial" if serial else "two streams", "eager   : %.4f ms/step wall, %.4f ms/step host enqueue" % timed(eager, n), flush=True)
cap = torch.cuda.Stream(device=dev)
side_stream(dev, cap)
with torch.cuda.stream(cap):
    for i in range(3):
        eng.train_step(s_seq, s_pos, N, 5e-4, **kw)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
    eng.train_step(s_seq, s_pos, N, 5e-4, **kw)
torch.cuda.synchronize()


def replay(i):
    s_seq.copy_(batches[i % 4][0])
    s_pos.copy_(batches[i % 4][1][:B])
    g.replay()


print(name, "replay  : %.4f ms/step wall, %.4f ms/step host enqueue" % timed(replay, n), "loss", float(eng.loss.item()), flush=True)
print(name, "eager   : %.4f ms/step wall, %.4f ms/step host enqueue" % timed(eager, n), flush=True)
