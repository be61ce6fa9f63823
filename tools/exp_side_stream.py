"""Which side streams make the step 3x slower?  One fresh engine per candidate stream (cfgY shape), the stream forced into
Engine._side before the first step.  Candidates: torch's high-priority pool in creation order, then its normal-priority pool."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ader_amd.engine import Engine  # noqa: E402

dev = torch.device("cuda:0")
label, N, B, E = bench.REAL_SHAPES["cfgY"]
T = 50
batches = [bench.synth_batch(B + E, T, N, 1000 * s + 77, dev, "realistic") for s in range(4)]
Np = int(0.9 * N)
teacher = torch.empty(E, (Np + 3) // 4 * 4, device=dev)[:, :Np]
teacher.copy_(torch.randn(E, Np, generator=torch.Generator().manual_seed(7)))
kw = dict(rate=0.3, teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device=dev), lambda_=0.8)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)


def run(side, tag):
    eng = Engine(N, maxlen=T, seed=0, device=dev)
    eng.pack_density = 0.1
    eng._side = side
    for i in range(12):
        eng.train_step(batches[i % 4][0], batches[i % 4][1][:B], N, 5e-4, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(400):
        eng.train_step(batches[i % 4][0], batches[i % 4][1][:B], N, 5e-4, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 400 * 1e3
    print("%-10s stream 0x%x id %s: %.4f ms" % (tag, eng._side.cuda_stream, getattr(eng._side, "stream_id", "?"), ms), flush=True)


for pr, tag in ((-1, "high"), (0, "normal")):
    for k in range(14):
        run(torch.cuda.Stream(device=dev, priority=pr), "%s[%d]" % (tag, k))
