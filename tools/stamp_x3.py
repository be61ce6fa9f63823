"""Per-segment clocks of k_lx3g (diagnostic build with -DG3_STAMP).  Dev tool."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd import _lib
from ader_amd.engine import Engine
from bench import synth_batch
N, B, T = 1_000_000, 512, 50
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device=dev, logits_dtype="x3")
batches = [synth_batch(B, T, N, 1000 * s, dev) for s in range(4)]
for i in range(8):
    eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 1024))()
lib.ader_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.ader_dbg_read(buf, 8 * 1024))
a = np.array(buf[:], dtype=np.float64).reshape(1024, 8)
a = a[a[:, 7] > 0][:512]
nb = a[:, 7]
names = ["loop/stamp", "barrier", "load issue", "S phase", "store (wait+cvt+write)", "tr issue + softmax", "O phase"]
per = a[:, :7] / nb[:, None]
tot = per.sum(1)
print("blocks per workgroup", nb.mean(), "clocks per block (median)", np.median(tot))
for k, n in enumerate(names):
    print("%-26s median %8.0f  share %5.1f %%" % (n, np.median(per[:, k]), 100 * np.median(per[:, k]) / np.median(tot)))
