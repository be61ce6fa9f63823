"""One side of tools/ab_f16x3.sh: the float32-grade logit forward (ader_lx3_fwd: operand cut + k_lx3p + merge) of whichever library
ADER_HIP_LIB names, at the headline shape (10^6 items, 512 rows, H = 150), against an fp64 reference computed on the GPU by torch:
log-sum-exp, loss and dRep errors, and the launcher's time.  Table values: `init` = Glorot range of a 1M-item table (+-0.00245),
`trained` = N(0, 0.05)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ader_amd import _lib
from ader_amd._lib import call, ptr

case = sys.argv[1] if len(sys.argv) > 1 else "init"
N, B, H = 1_000_000, 512, 150
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(3)
rep = torch.randn(B, H, generator=g).to(dev)
emb = torch.zeros(N + 1, H)
emb[1:] = (torch.rand(N, H, generator=g) * 2 - 1) * 0.00245 if case == "init" else torch.randn(N, H, generator=g) * 0.05
emb = emb.to(dev)
pos = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int32).to(dev)
Bp = (B + 127) // 128 * 128
st = torch.cuda.current_stream().cuda_stream
i32, f32 = dict(dtype=torch.int32, device=dev), dict(dtype=torch.float32, device=dev)
lab, ncol, trow = torch.zeros(Bp, **i32), torch.zeros(Bp, **i32), torch.zeros(Bp, **i32)
wrow = torch.zeros(Bp, **f32)
call("ader_build_rowinfo", ptr(pos), B, None, None, 0, N, 0, 1.0 / B, 0.0, Bp, ptr(lab), ptr(ncol), ptr(wrow), ptr(trow), st)
R = call("ader_lbf_ranges", N, Bp)
rep_hi = torch.zeros(Bp * 168, dtype=torch.bfloat16, device=dev)
rep_lo = torch.zeros(Bp * 168, dtype=torch.bfloat16, device=dev)
pm, pl, pO = torch.empty(R * Bp, **f32), torch.empty(R * Bp, **f32), torch.empty(R * Bp * 160, **f32)
lse, off, rowloss, loss, drep = (torch.empty(Bp, **f32), torch.empty(Bp, **f32), torch.empty(Bp, **f32), torch.zeros(1, **f32),
                                 torch.zeros(B, H, **f32))


def fwd():
    call("ader_lx3_fwd", ptr(rep), ptr(emb), N, B, Bp, H, N, ptr(lab), ptr(wrow), ptr(rep_hi), ptr(rep_lo), ptr(pm), ptr(pl), ptr(pO),
         ptr(lse), ptr(off), ptr(rowloss), ptr(loss), ptr(drep), st)


for _ in range(3):
    fwd()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    fwd()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
# fp64 reference in row chunks (64 rows x 10^6 items x 8 B = 0.5 GB per chunk)
E64 = emb[1:].double()
lse_ref = torch.empty(B, dtype=torch.float64, device=dev)
drep_ref = torch.empty(B, H, dtype=torch.float64, device=dev)
tgt = torch.empty(B, dtype=torch.float64, device=dev)
for s in range(0, B, 64):
    lg = rep[s:s + 64].double() @ E64.T
    z = torch.logsumexp(lg, dim=1)
    p = torch.exp(lg - z[:, None])
    idx = pos[s:s + 64].long() - 1
    tgt[s:s + 64] = lg.gather(1, idx[:, None])[:, 0]
    lse_ref[s:s + 64] = z
    drep_ref[s:s + 64] = (p @ E64 - E64[idx]) / B
loss_ref = float(((lse_ref - tgt) / B).sum())
e_lse = (lse[:B].double() - lse_ref).abs()
e_dr = (drep.double() - drep_ref).abs() / drep_ref.abs().max()
print("%-8s %-52s lse |err| max %.3e rms %.3e   dRep err/max|dRep| max %.3e rms %.3e   loss %.9f (fp64 %.9f, |d| %.2e)   %.4f ms" % (
    case, os.path.basename(_lib.LIB_PATH), float(e_lse.max()), float(e_lse.pow(2).mean().sqrt()), float(e_dr.max()),
    float(e_dr.pow(2).mean().sqrt()), float(loss), loss_ref, abs(float(loss) - loss_ref), ms))
