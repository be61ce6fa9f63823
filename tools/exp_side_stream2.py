"""Follow-up to exp_side_stream.py: does a ping-pong micro-benchmark (main kernel -> side waits -> side kernel -> main waits) see the
slow streams, are they stable within a process, and do streams made by hipStreamCreateWithPriority behave the same?"""
import ctypes
import os
import sys
import time

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
x = torch.zeros(1 << 16, device=dev)
y = torch.zeros(1 << 16, device=dev)


def pingpong(side, n=300):
    main = torch.cuda.current_stream()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            x.add_(1.0)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                y.add_(1.0)
            main.wait_stream(side)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def step_ms(side):
    eng = Engine(N, maxlen=T, seed=0, device=dev)
    eng.pack_density = 0.1
    eng._side = side
    for i in range(12):
        eng.train_step(batches[i % 4][0], batches[i % 4][1][:B], N, 5e-4, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300):
        eng.train_step(batches[i % 4][0], batches[i % 4][1][:B], N, 5e-4, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 300 * 1e3


hip = ctypes.CDLL("libamdhip64.so")
streams = []
for pr, tag in ((-1, "high"), (0, "normal")):
    for k in range(10):
        streams.append(("%s[%d]" % (tag, k), torch.cuda.Stream(device=dev, priority=pr)))
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("priority range least %d greatest %d" % (lo.value, hi.value), flush=True)
for pr in (hi.value, 0):
    for k in range(6):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, pr) == 0          # hipStreamNonBlocking
        streams.append(("own%+d[%d]" % (pr, k), torch.cuda.ExternalStream(h.value, device=dev)))
for tag, s in streams:
    pp = pingpong(s)
    a = step_ms(s)
    b = step_ms(s)
    print("%-11s ping-pong %6.1f us | step %.4f %.4f ms" % (tag, pp, a, b), flush=True)
