"""Per-phase clocks of k_seq_pack_plan (diagnostic build with -DPLAN_STAMP: ADER_HIP_LIB=ader_amd/variants/libader_hip_planstamp.so).  Dev tool."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd import _lib
from ader_amd.engine import Engine
from bench import synth_batch
N, B, T = 25750, 614, 50
eng = Engine(N, maxlen=T)
sq, _ = synth_batch(B, T, N, 1, "cuda", "realistic")
for _ in range(5):
    pk = eng._pack_plan(sq, "tst")
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
lib.ader_dbg_read_plan.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.ader_dbg_read_plan(buf, 16))
names = ["lengths of 64 sessions + ticket (last workgroup)", "slen read-back + init + class sums", "four block scans", "assignment loop (atomics)",
         "compaction scan + tile rows", "per-session first row + hdr"]
a = np.array(buf[:6], dtype=np.float64)
for n_, v in zip(names, a):
    print("%-45s %8.0f clocks %5.1f %%" % (n_, v, 100 * v / a.sum()))
print("sum %.0f clocks; hdr" % a.sum(), pk["hdr"].cpu().numpy()[:4])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    eng._pack_plan(sq, "tst")
e1.record(); torch.cuda.synchronize()
print("back-to-back launches: %.1f us each" % (e0.elapsed_time(e1) * 1000 / 200))
