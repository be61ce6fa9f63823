"""Per-segment clocks of k_tabp's three wave roles (diagnostic build with -DTP_STAMP: ADER_HIP_LIB=ader_amd/variants/libader_hip_stamp.so).  Dev tool."""
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
n = 16 * 256
buf = (ctypes.c_ulonglong * n)()
lib.ader_dbg_read_tp.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.ader_dbg_read_tp(buf, n))
a = np.array(buf[:], dtype=np.float64).reshape(256, 16)
names = ["GEMM barrier wait", "GEMM S phase (+ slot-0 exp)", "GEMM dE phase (+ in-gap exp)", "GEMM hand-off (x loads, H1, F write, H2, cut)",
         "LOADER barrier wait", "LOADER put (wait + ds_write)", "LOADER fetch issue", "-",
         "ADAM barrier wait", "ADAM rounds + issue-ahead", "ADAM slot 0 (sparse rows)", "ADAM H1 + H2"]
for r0, nm in ((0, "GEMM wave 0"), (4, "LOADER wave 4"), (8, "ADAM wave 6")):
    tot = a[:, r0:r0 + 4].sum(1)
    print("%s: clocks per launch (median over workgroups) %.0f" % (nm, np.median(tot)))
    for k in range(4):
        print("   %-50s median %9.0f  share %5.1f %%" % (names[r0 + k], np.median(a[:, r0 + k]), 100 * np.median(a[:, r0 + k]) / np.median(tot)))
