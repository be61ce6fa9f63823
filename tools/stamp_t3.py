"""Per-segment clocks of k_tab16x3 (diagnostic build with -DT3_STAMP).  Dev tool."""
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
n = 12 * 976
buf = (ctypes.c_ulonglong * n)()
lib.ader_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.ader_dbg_read(buf, n))
a = np.array(buf[:], dtype=np.float64).reshape(-1, 12)
a = a[a.sum(1) > 0]
names = ["theta load + LDS write", "barrier 1", "sparse prefetch + cut", "barrier 2", "chunk: S-phase tail.. (16x) / pre-opt",
         "chunk: vmcnt + barrier (16x)", "chunk: DMA issue + S phase (16x)", "chunk: tr issue + exp (16x)", "dE phase 16x + round load + staging",
         "sparse terms + barrier", "Adam rounds + stores"]
tot = a[:, :11].sum(1)
print("tiles sampled", len(a), "clocks per tile (median)", np.median(tot))
for k, nme in enumerate(names):
    print("%-44s median %8.0f  share %5.1f %%" % (nme, np.median(a[:, k]), 100 * np.median(a[:, k]) / np.median(tot)))
