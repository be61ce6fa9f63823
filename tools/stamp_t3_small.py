"""Per-segment clocks of k_tab32x3 at a real-data step shape (diagnostic build with -DT3_STAMP: ADER_HIP_LIB=ader_amd/variants/
libader_hip_t3stamp.so).  usage: python tools/stamp_t3_small.py [cfgY|cfgD] [uniform]   (uniform: ids ~ U[1,N] instead of Zipf)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ader_amd import _lib
from ader_amd.engine import Engine
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfgY"
uniform = "uniform" in sys.argv
_, N, B, E = bench.REAL_SHAPES[name]
T = 50
dev = torch.device("cuda", 0)
batches = [bench.synth_batch(B + E, T, N, 1000 * s + 77, "cpu", "realistic") for s in range(4)]
if uniform:
    rs = np.random.RandomState(5)
    nb = []
    for sq, ps in batches:
        sq, ps = sq.numpy().copy(), ps.numpy().copy()
        sq[sq != 0] = rs.randint(1, N + 1, size=int((sq != 0).sum()))
        ps[:] = rs.randint(1, N + 1, size=len(ps))
        nb.append((torch.from_numpy(sq), torch.from_numpy(ps)))
    batches = nb
batches = [(a.to(dev), b.to(dev)) for a, b in batches]
eng = Engine(N, maxlen=T, seed=0, device=dev)
eng.pack_density = 0.1
Np = int(0.9 * N)
teacher = torch.empty(E, (Np + 3) // 4 * 4, device=dev)[:, :Np]
teacher.copy_(torch.randn(E, Np, generator=torch.Generator().manual_seed(7)))
kw = dict(rate=0.3, teacher=teacher, ex_trow=torch.arange(E, dtype=torch.int32, device=dev), lambda_=0.8)
for i in range(8):
    eng.train_step(batches[i % 4][0], batches[i % 4][1][:B], N, 5e-4, **kw)
torch.cuda.synchronize()
lib = _lib.load()
n = 12 * 1024
buf = (ctypes.c_ulonglong * n)()
lib.ader_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.ader_dbg_read(buf, n))
a = np.array(buf[:], dtype=np.float64).reshape(-1, 12)
a = a[a.sum(1) > 0]
names = ["0 prologue: operand rows of the pair cut to hi/lo", "1 barrier", "2 sparse prefetch + setup", "3 -", "4 chunk head (+ pre-optimiser)",
         "5 chunk: vmcnt + barrier", "6 chunk: DMA issue + S phase", "7 chunk: transposed reads + exp + dE MFMAs", "8 optimiser: m/v loads, F tile, barriers",
         "9 sparse terms", "10 theta load + Adam + stores"]
tot = a[:, :11].sum(1)
print(name, "uniform ids" if uniform else "zipf ids", "| workgroups sampled", len(a), "| 100 MHz clocks per workgroup: median %.0f  max %.0f  (x10 ns)" % (np.median(tot), tot.max()))
for k, nme in enumerate(names):
    print("%-52s median %8.0f  max %8.0f  share of median %5.1f %%" % (nme, np.median(a[:, k]), a[:, k].max(), 100 * np.median(a[:, k]) / np.median(tot)))
