"""Times Engine.forward (training mode) with the one-launch kernel and with the per-op chain.  Dev tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd.engine import Engine
B, T, N = 512, 50, 100000
eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0)
rs = np.random.RandomState(0)
seq = eng._dev_i32(rs.randint(1, N + 1, size=(B, T)))
for fused in (True, False, True):
    eng.seq_fused = fused
    for _ in range(3):
        eng.forward(seq, training=True, rate=0.3, step=1, save=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        eng.forward(seq, training=True, rate=0.3, step=1, save=True)
    b.record(); torch.cuda.synchronize()
    print("fused" if fused else "per-op", "%.1f us" % (a.elapsed_time(b) / 20 * 1e3))
