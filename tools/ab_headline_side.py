"""Headline step (cfg-S: 1M items, B=512) per side-stream candidate and with / without the fused final-LayerNorm backward, one
process, alternating (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd.engine import Engine, side_stream
from bench import synth_batch

N, B, T = 1_000_000, 512, 50
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device=dev, logits_dtype="x3")
batches = [synth_batch(B, T, N, 1000 * s, dev) for s in range(4)]


def run(steps=60):
    for i in range(6):
        eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


picked = side_stream(dev, torch.cuda.current_stream())
cands = [("picked", picked)] + [("high[%d]" % k, torch.cuda.Stream(device=dev, priority=-1)) for k in range(4)] \
    + [("normal[%d]" % k, torch.cuda.Stream(device=dev, priority=0)) for k in range(4)]
for rep in range(3):
    for tag, s in cands:
        eng._side = s
        print(rep, "%-10s" % tag, "%.4f" % run(), flush=True)
eng._side = picked
for rep in range(4):
    for f in (True, False):
        eng.fuse_final_ln = f
        print(rep, "fuse_final_ln=%s" % f, "%.4f" % run(), flush=True)
