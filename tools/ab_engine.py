"""A/B of engine scheduling switches on the bench workload, alternating runs in one process (dev tool).

python tools/ab_engine.py [--dtype x3|bf16] [--reps 3] name:attr=val[,attr=val] ...     (values: python literals)"""
import argparse, ast, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd.engine import Engine
from bench import synth_batch

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="x3")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("cfgs", nargs="*")
args = ap.parse_args()

N, B, T = 1_000_000, 512, 50
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device=dev, logits_dtype=args.dtype)
batches = [synth_batch(B, T, N, 1000 * s, dev) for s in range(4)]


def run(steps):
    for i in range(6):
        eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


cfgs = []
for c in args.cfgs or ["default:"]:
    name, _, kvs = c.partition(":")
    kv = {}
    for item in filter(None, kvs.split(",")):
        k, _, v = item.partition("=")
        kv[k] = ast.literal_eval(v)
    cfgs.append((name, kv))
base = {k: getattr(eng, k) for _, kv in cfgs for k in kv}
for rep in range(args.reps):
    for name, kv in cfgs:
        for k, v in base.items():
            setattr(eng, k, v)
        for k, v in kv.items():
            setattr(eng, k, v)
        print(rep, name, round(run(args.steps), 4), flush=True)
