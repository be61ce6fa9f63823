"""Per-phase clocks of k_seq_fwd (waves 0 and 9 of each workgroup; diagnostic build with -DSF_STAMP:
ADER_HIP_LIB=ader_amd/variants/libader_hip_sfstamp.so).  Dev tool."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd import _lib
from ader_amd.engine import Engine
from bench import synth_batch
N, B, T = 1_000_000, 1024, 50
dev = torch.device("cuda", 0)
eng = Engine(N, maxlen=T, hidden_units=150, num_blocks=2, num_heads=1, seed=0, device=dev, logits_dtype="x3")
batches = [synth_batch(B, T, N, 1000 * s, dev) for s in range(4)]
for i in range(8):
    eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
torch.cuda.synchronize()
lib = _lib.load()
n = 2 * 40 * 1024
buf = (ctypes.c_ulonglong * n)()
lib.ader_dbg_read_sf.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.ader_dbg_read_sf(buf, n))
a = np.array(buf[:], dtype=np.float64).reshape(1024, 2, 40)
blk = ["LN1 (+ small params)", "barrier after LN1", "Q mma + next B-frag issue", "Q barrier + epilogue", "K mma + issue", "K epilogue",
       "V mma", "V barrier + epilogue", "__syncthreads", "residual loads issue", "scores + softmax + P tile (3 barriers)", "barrier before P.V",
       "P.V mma + W1 issue + x1 epilogue", "barrier + LN2", "barrier + FFN1 mma + epilogue", "barrier + FFN2 mma + epilogue (+ last barrier)"]
names = ["prologue (gather + barrier)"] + ["block 0: " + x for x in blk] + ["block 1 (pruned): " + x for x in blk] + ["final LN wait", "final LN"]
idx = list(range(0, 33)) + [33, 34]
for w, nm in ((0, "wave 0"), (1, "wave 9")):
    tot = a[:, w, :].sum(1)
    print("%s: clocks per workgroup, median %.0f (s_memtime ticks at 100 MHz x ? -- relative shares are what matters)" % (nm, np.median(tot)))
    for k, name in zip(idx, names):
        print("   %-75s median %8.0f  share %5.1f %%" % (name, np.median(a[:, w, k]), 100 * np.median(a[:, w, k]) / np.median(tot)))
