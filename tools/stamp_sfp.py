"""Per-phase clocks of k_seqp_fwd (waves 0 and 4 of each workgroup; diagnostic build with -DSFP_STAMP:
ADER_HIP_LIB=ader_amd/variants/libader_hip_sfpstamp.so).  Dev tool.  usage: stamp_sfp.py [cfgY|cfgD]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd import _lib
from ader_amd.engine import Engine
from bench import synth_batch
w = sys.argv[1] if len(sys.argv) > 1 else "cfgY"
N, B = {"cfgY": (25750, 614), "cfgD": (43105, 399)}[w]
T = 50
eng = Engine(N, maxlen=T)
eng.pack_sessions = True
if len(sys.argv) > 2:
    eng.pack_window = tuple(int(v) for v in sys.argv[2].split(","))
batches = [synth_batch(B, T, N, 1000 * s, "cuda", "realistic") for s in range(4)]
for i in range(8):
    eng.train_step(*batches[i % 4], N, 5e-4, rate=0.3)
torch.cuda.synchronize()
nt = int(eng._act["pack"]["hdr"][0].item())
tr = eng._act["pack"]["trows"][:nt].cpu().numpy()
print("tiles", nt, "rows per tile: min %d median %d max %d" % (tr.min(), np.median(tr), tr.max()))
lib = _lib.load()
n = 2 * 40 * 1024
buf = (ctypes.c_ulonglong * n)()
lib.ader_dbg_read_sfp.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.ader_dbg_read_sfp(buf, n))
a = np.array(buf[:], dtype=np.float64).reshape(1024, 2, 40)[:nt]
blk = ["LN1 (+ small params)", "barrier after LN1", "Q mma + next B-frag issue", "Q barrier + epilogue", "K mma + issue", "K epilogue",
       "V mma", "V barrier + epilogue", "__syncthreads", "residual loads issue", "scores + softmax + P tile (3 barriers)", "barrier before P.V",
       "P.V mma + W1 issue + x1 epilogue", "barrier + LN2", "barrier + FFN1 mma + epilogue", "barrier + FFN2 mma + epilogue (+ last barrier)"]
names = ["prologue (gather + barrier)"] + ["block 0: " + x for x in blk] + ["block 1 (pruned): " + x for x in blk] + ["final LN wait", "final LN"]
idx = list(range(0, 33)) + [33, 34]
for wv, nm in ((0, "wave 0"), (1, "wave 4")):
    tot = a[:, wv, :].sum(1)
    print("%s: clocks per workgroup, median %.0f" % (nm, np.median(tot)))
    for k, name in zip(idx, names):
        print("   %-75s median %8.0f  share %5.1f %%" % (name, np.median(a[:, wv, k]), 100 * np.median(a[:, wv, k]) / np.median(tot)))
