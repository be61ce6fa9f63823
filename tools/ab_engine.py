"""A/B of engine scheduling switches on the bench workload, alternating runs in one process (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd.engine import Engine
from bench import synth_batch

N, B, T = 1_000_000, 512, 50
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device=dev, logits_dtype="bf16")
batches = [synth_batch(B, T, N, 1000 * s, dev) for s in range(4)]


def run(steps=40):
    for i in range(6):
        eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


cfgs = [("default", {}), ("late_main", {"late_side_stream": False}), ("lists_main", {"lists_side_stream": False}),
        ("both_main", {"late_side_stream": False, "lists_side_stream": False})]
for rep in range(3):
    for name, kv in cfgs:
        eng.late_side_stream, eng.lists_side_stream = True, True
        for k, v in kv.items():
            setattr(eng, k, v)
        print(rep, name, round(run(), 4), flush=True)
